#!/usr/bin/env python3
"""Kernel time of consecutive find_matches calls from a cold GPU: how long the clock ramp lasts (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth
from MTM import _lib
img, units, plants = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.Context(0)
ctx.set_image(img); ctx.set_templates([(u[1], None) for u in units], 5)
time.sleep(2.0)
ts, wall = [], []
t0 = time.perf_counter()
for i in range(600):
    ctx.find_matches(0, 0.5); ts.append(ctx.timing()["ncc_kernel_ms"]); wall.append(time.perf_counter() - t0)
for a in range(0, 600, 40):
    print("calls %3d-%3d (t=%.0f ms): ncc median %.4f ms" % (a, a + 39, wall[a] * 1e3, float(np.median(ts[a:a + 40]))), flush=True)
