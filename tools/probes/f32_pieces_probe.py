#!/usr/bin/env python3
"""Round 6: the one-product screen of the float32 hits-only routes (MTM_OPT_F32_MFMA = 1, mtm_timing.f32_pieces) against the
three-product screen (option value 3) and the float64 kernel (0): per-call time, score-kernel time, pieces / route taken and
whether the records are identical - 4K x 32 templates 64x64 and 1080p x 8, the synthetic (sparse) image and a photograph-like
one, the normalised methods with a threshold, N_object == 1, the raw sums with a threshold, masked templates (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth
from MTM import _lib

def timed(ctx, tl, im, method, mode, thr, reps=5):
    ctx.search(tl, im, method, mode, thr)
    ts, r = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = ctx.search(tl, im, method, mode, thr).copy()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts)), r, ctx.timing()

ctxs = {}
for name, opt in (("one-product tier", 1), ("three products", 3), ("float64 kernel", 0)):
    c = _lib.Context(0)
    c.set_option(_lib.OPT_F32_MFMA, opt)
    ctxs[name] = c
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for cfg in ("cfg2", "cfg3_32"):
    img, units, plants = synth.make_config(cfg)
    f = img.astype(np.float32) * 0.731 + 3.25
    tl = [(u[1].astype(np.float32) * 0.731 + 3.25, None) for u in units]
    disc = (np.hypot(*np.mgrid[-31.5:32, -31.5:32]) <= 30).astype(np.float32)
    tlm = [(t, disc) for t, _ in tl]
    smooth = synth.smooth_u8(5, img.shape).astype(np.float32) * 0.731 + 3.25
    cases = [("sparse image, TM_CCOEFF_NORMED thr 0.5", f, tl, 5, _lib.PEAKS_LOCAL, 0.5),
             ("sparse image, TM_CCORR_NORMED thr 0.95", f, tl, 3, _lib.PEAKS_LOCAL, 0.95),
             ("sparse image, TM_SQDIFF_NORMED thr 0.1", f, tl, 1, _lib.PEAKS_LOCAL, 0.1),
             ("sparse image, TM_CCOEFF_NORMED N_object == 1", f, tl, 5, _lib.PEAKS_GLOBAL, 0.0),
             ("sparse image, TM_CCOEFF (raw) N_object == 1", f, tl, 4, _lib.PEAKS_GLOBAL, 0.0),
             ("sparse image, TM_CCOEFF (raw) thr", f, tl, 4, _lib.PEAKS_LOCAL, None),
             ("sparse image, masked TM_CCORR_NORMED thr 0.9", f, tlm, 3, _lib.PEAKS_LOCAL, 0.9),
             ("sparse image, masked TM_CCORR_NORMED N_object == 1", f, tlm, 3, _lib.PEAKS_GLOBAL, 0.0),
             ("photograph-like image, TM_CCOEFF_NORMED thr 0.5", smooth, tl, 5, _lib.PEAKS_LOCAL, 0.5),
             ("photograph-like image, TM_CCOEFF_NORMED thr 0.9", smooth, tl, 5, _lib.PEAKS_LOCAL, 0.9)]
    for label, im, tls, method, mode, thr in cases:
        if thr is None:         # raw sums: a threshold half way up to the planted copies' score
            t0 = tls[0][0].astype(np.float64)
            thr = 0.5 * float(((t0 - t0.mean()) ** 2).sum())
        ref = None
        for name, c in ctxs.items():
            if name == "float64 kernel" and (quick or "photograph" in label) and cfg != "cfg2":
                continue
            ms, r, tm = timed(c, tls, im, method, mode, thr, reps=3 if name == "float64 kernel" else 5)
            if ref is None:
                ref = r
            same = "identical" if r.tobytes() == ref.tobytes() else "DIFFERENT (%d vs %d)" % (len(r), len(ref))
            print("%-7s %-52s %-17s %8.2f ms  kernel %6.2f ms  route %d pieces %d  %5d records %s" % (
                cfg, label, name, ms, tm["ncc_kernel_ms"], tm["f32_route"], tm["f32_pieces"], len(r), same), flush=True)
