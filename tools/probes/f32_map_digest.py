#!/usr/bin/env python3
"""Digest of float64-kernel score maps of float32 images (MTM_OPT_F32_MFMA = 0) and of masked float32 hit lists - run once per
library build (MTM_LIB_PATH=...) and compare the lines: a change that must leave the float32 window statistics bit for bit
what they were (round 6: hsum_lds_kernel) shows here if it does not (GPU box)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
from MTM import _lib
rng = np.random.default_rng(3)
ctx = _lib.Context(0)
ctx.set_option(_lib.OPT_F32_MFMA, 0)
for (H, W, h, w) in ((300, 420, 32, 40), (2160, 3840, 64, 64), (517, 4333, 9, 130), (260, 400, 24, 24), (700, 4200, 100, 300)):
    im = (rng.normal(40.0, 11.0, (H, W)) + np.linspace(0, 300, W)[None, :]).astype(np.float32)
    t = np.ascontiguousarray(im[5:5 + h, 7:7 + w])
    for method in (5, 3, 1, 0):
        ctx.set_image(im)
        ctx.set_templates([(t, None)], method)
        m = ctx.score_map(0, (H - h + 1, W - w + 1))
        print("%dx%d %dx%d m%d %s" % (H, W, h, w, method, hashlib.sha256(m.tobytes()).hexdigest()[:16]), flush=True)
    if w <= 256:
        disc = (np.hypot(*np.mgrid[-(h - 1) / 2:(h + 1) / 2, -(w - 1) / 2:(w + 1) / 2]) <= min(h, w) / 2).astype(np.float32)
        c2 = _lib.Context(0)
        r = c2.search([(t, disc)], im, 3, _lib.PEAKS_LOCAL, 0.9)
        print("%dx%d %dx%d masked m3 %s (%d records, route %d)" % (H, W, h, w, hashlib.sha256(r.tobytes()).hexdigest()[:16], len(r), c2.timing()["f32_route"]), flush=True)
        c2.close()
