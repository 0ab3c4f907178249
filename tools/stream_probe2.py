#!/usr/bin/env python3
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth, MTM
from MTM import _lib
img, units, plants = synth.make_config("cfg3_32")
ctx = _lib.Context(0)
ctx.set_image(img); ctx.set_templates([(u[1], None) for u in units], 5)
frames = [np.ascontiguousarray(np.roll(img, 64 * k, axis=1)) for k in range(4)] * 6
for rep in range(3):
    t0 = time.perf_counter()
    for f in frames: ctx.find_matches(0, 0.5)
    t1 = time.perf_counter()
    for f in frames: ctx.find_matches(0, 0.5, next_image=f)
    t2 = time.perf_counter()
    tm = ctx.timing()
    for f in frames:
        ctx.set_image(f); ctx.find_matches(0, 0.5)
    t3 = time.perf_counter()
    n = len(frames)
    print("find %.3f | find_next %.3f | set_image+find %.3f ms  (gpu total %.3f ncc %.3f)" % ((t1 - t0) / n * 1e3, (t2 - t1) / n * 1e3, (t3 - t2) / n * 1e3, tm["total_ms"], tm["ncc_kernel_ms"]), flush=True)
