#!/usr/bin/env python3
"""Masked float32 templates: the bf16 screen + exact re-scoring route (MTM_OPT_F32_MFMA = 1, mtm_maskf32.hip.h) against the
float64 kernel (MTM_OPT_F32_MFMA = 0) - records must be identical - and its timing at 4K x 32 templates (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
from MTM import _lib

rng = np.random.default_rng(11)


def disc(h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    return ((((yy - h / 2 + 0.5) / (h / 2)) ** 2 + ((xx - w / 2 + 0.5) / (w / 2)) ** 2) <= 1.0).astype(np.float32)


def case(H, W, h, w, n, kind, scale):
    img = (rng.random((H, W)).astype(np.float32) * scale).astype(np.float32)
    if kind == "step":
        img[:, W // 2:] += np.float32(scale)
    units = []
    for i in range(n):
        y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
        t = img[y:y + h, x:x + w].copy()
        if i % 3 == 1:
            t = (t + rng.normal(0, 0.05 * scale, t.shape)).astype(np.float32)
        m = disc(h, w) if i % 2 == 0 else rng.random((h, w)).astype(np.float32)
        units.append((t, m))
    return img, units


def run(ctx, img, units, method, thr):
    return ctx.search(units, img, method, _lib.PEAKS_LOCAL, thr).copy()


fast, exact = _lib.Context(0), _lib.Context(0)
exact.set_option(_lib.OPT_F32_MFMA, 0)
bad = 0
for (H, W, h, w, n, kind, scale) in [(150, 333, 24, 40, 5, "noise", 1.0), (150, 333, 20, 70, 20, "noise", 255.0), (300, 420, 33, 17, 7, "step", 1.0),
                                    (260, 610, 64, 64, 37, "noise", 65535.0), (200, 300, 9, 130, 3, "step", 100.0)]:
    img, units = case(H, W, h, w, n, kind, scale)
    for method in (3, 0):
        if method == 3:
            thrs = (0.9, 0.5, 0.999)
        else:
            base = float(np.median([(units[0][0] * units[0][1]).astype(np.float64).var() * h * w]))
            thrs = (1e-3 * base + 1e-6, 0.3 * base)
        for thr in thrs:
            a = run(fast, img, units, method, thr)
            ta = fast.timing()
            b = run(exact, img, units, method, thr)
            same = a.tobytes() == b.tobytes()
            bad += 0 if same else 1
            print("%-5s %4dx%-4d %3dx%-3d n=%2d m%d thr %-10.4g: %5d records, %s  (route %s kernel %s)" % (
                kind, H, W, h, w, n, method, thr, len(b), "identical" if same else "DIFFERENT (%d vs %d)" % (len(a), len(b)),
                ta["f32_route"], ta["kernel_used"]), flush=True)
            if thr == thrs[0]:                       # N_object == 1: the screen with the templates' own best lower bound
                a = fast.search(units, img, method, _lib.PEAKS_GLOBAL, 0.0).copy()
                ta = fast.timing()
                b = exact.search(units, img, method, _lib.PEAKS_GLOBAL, 0.0).copy()
                same = a.tobytes() == b.tobytes()
                bad += 0 if same else 1
                print("      N_object == 1 m%d: %d records, %s (route %s)" % (method, len(b), "identical" if same else "DIFFERENT", ta["f32_route"]), flush=True)
print("cases with different records:", bad)
# timing: 4K x 32 templates 64x64, disc masks, TM_CCORR_NORMED
H, W = 2160, 3840
img = rng.random((H, W)).astype(np.float32)
units = []
for i in range(32):
    y, x = int(rng.integers(0, H - 64)), int(rng.integers(0, W - 64))
    units.append((img[y:y + 64, x:x + 64].copy(), disc(64, 64)))
for name, ctx in (("bf16 screen + exact re-scoring", fast), ("float64 kernel", exact)):
    for _ in range(2):
        r = run(ctx, img, units, 3, 0.9)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = run(ctx, img, units, 3, 0.9)
        ts.append(time.perf_counter() - t0)
    print("4K x 32 masked float32, TM_CCORR_NORMED: %-32s %.2f ms per call, %d records, route %s" % (name, 1e3 * float(np.median(ts)), len(r), ctx.timing()["f32_route"]), flush=True)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        r = ctx.search(units, img, 3, _lib.PEAKS_GLOBAL, 0.0)
        ts.append(time.perf_counter() - t0)
    print("    ... N_object == 1: %.2f ms per call, route %s" % (1e3 * float(np.median(ts[1:])), ctx.timing()["f32_route"]), flush=True)
