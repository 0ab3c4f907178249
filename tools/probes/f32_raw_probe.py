#!/usr/bin/env python3
"""float32 raw-sum methods (TM_SQDIFF 0, TM_CCORR 2, TM_CCOEFF 4) with a threshold, local extrema: the bf16 kernel listing
by the bound of the sum (round 5, route 1) against the float64 kernel (MTM_OPT_F32_MFMA = 0), 1080p x 8 and 4K x 32 (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth
from MTM import _lib
for cfg in ("cfg2", "cfg3_32"):
    img, units, plants = synth.make_config(cfg)
    im = img.astype(np.float32) * np.float32(1.0 / 255.0)
    tl = [(u[1].astype(np.float32) * np.float32(1.0 / 255.0), None) for u in units]
    for method in (0, 2, 4):
        fast, exact = _lib.Context(0), _lib.Context(0)           # (an overflow leaves a context in its back-off for some calls)
        exact.set_option(_lib.OPT_F32_MFMA, 0)
        # the threshold: between the planted copies' scores and everything else's, from the float64 kernel's global extrema
        g = exact.search(tl, im, method, _lib.PEAKS_GLOBAL, 0.0)["score"].astype(np.float64)
        if method == 0:
            thr = max(1.0, float(g.max()) * 4.0)
        elif method == 2:
            thr = float(g.min()) * 0.98
        else:
            thr = float(g.min()) * 0.5
        out = {}
        for name, ctx in (("bf16+rescore", fast), ("float64", exact)):
            ctx.search(tl, im, method, _lib.PEAKS_LOCAL, thr)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); h = ctx.search(tl, im, method, _lib.PEAKS_LOCAL, thr); ts.append((time.perf_counter() - t0) * 1e3)
            out[name] = (sorted(ts)[2], ctx.timing()["f32_route"], h)
        same = out["bf16+rescore"][2].tobytes() == out["float64"][2].tobytes()
        print("%s method %d thr %.4g: bf16+rescore %.2f ms (route %d), float64 kernel %.2f ms, hits %d, identical %s" % (
            cfg, method, thr, out["bf16+rescore"][0], out["bf16+rescore"][1], out["float64"][0], len(out["float64"][2]), same), flush=True)
        fast.close(); exact.close()
