#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth
from MTM import _lib
img, units, plants = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.Context(0)
ctx.set_image(img); ctx.set_templates([(u[1], None) for u in units], 5)
def run(n=12):
    ts = []
    for i in range(n):
        ctx.find_matches(0, 0.5); ts.append(ctx.timing()["ncc_kernel_ms"])
    return min(ts), float(np.median(ts)), ctx.timing()["hits_only"]
print("initial (env MTM_HITS_ONLY=%s):" % os.environ.get("MTM_HITS_ONLY"), run())
for rnd in range(3):
    for mode in (0, 1):
        ctx.set_option(6, mode)
        print("set_option hits_only=%d ->" % mode, run(), flush=True)
