#!/usr/bin/env python3
"""Packed K (MTM_KPACK): score-kernel time for template widths that are not multiples of 64, 4K image, 32 templates
(plain tiling) and 8 templates (row-multiplexed).  Run twice: MTM_KPACK=0 / 1.  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth
from MTM import _lib
img = synth.rand_u8(3, 0, (2160, 3840))
rng = np.random.default_rng(2)
ctx = _lib.Context(0)
ctx.set_image(img)
for n in (32, 8):
    for (h, w) in ((64, 64), (41, 41), (48, 48), (59, 65), (80, 80), (100, 100), (32, 32), (24, 24), (16, 16)):
        tl = []
        for i in range(n):
            y, x = int(rng.integers(0, 2160 - h)), int(rng.integers(0, 3840 - w))
            tl.append((np.ascontiguousarray(img[y:y + h, x:x + w]), None))
        ctx.set_templates(tl, 5)
        for _ in range(30): ctx.find_matches(0, 0.5)
        ks = []
        for _ in range(30):
            hits = ctx.find_matches(0, 0.5); ks.append(ctx.timing()["ncc_kernel_ms"])
        print("KPACK=%s n=%2d %3dx%-3d kernel %.4f ms  hits %d" % (os.environ.get("MTM_KPACK", "1"), n, h, w, float(np.median(ks)), len(hits)), flush=True)
