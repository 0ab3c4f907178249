#!/usr/bin/env python3
"""Device-memory stability: many uploads of varying sizes, template sets, searches on one context and on
short-lived contexts; rocm-smi memory use before / after (GPU box)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
from MTM import _lib

def used_mb():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        p = line.split(",")
        if len(p) >= 3 and p[0].startswith("card"):
            return int(p[2]) / 1e6
    return float("nan")

rng = np.random.default_rng(0)
ctx = _lib.Context(0)
print("start: %.0f MB" % used_mb(), flush=True)
for rnd in range(3):
    t0 = time.time()
    for it in range(300):
        H, W = int(rng.integers(200, 1200)), int(rng.integers(200, 1600))
        img = rng.integers(0, 256, (H, W), dtype=np.uint8)
        n = int(rng.integers(1, 40))
        s = int(rng.integers(8, 64))
        tl = [(img[i:i + s, 2 * i:2 * i + s].copy(), None) for i in range(n)]
        ctx.set_image(img); ctx.set_templates(tl, 5)
        ctx.find_matches(0, 0.6)
        if it % 50 == 0:
            c2 = _lib.Context(0); c2.set_image(img); c2.set_templates(tl, 3); c2.find_matches(0, 0.9); del c2
    print("round %d: %.0f MB used, %.1f s" % (rnd, used_mb(), time.time() - t0), flush=True)
