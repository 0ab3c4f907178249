#!/usr/bin/env python3
"""Wall-clock breakdown of one bench step (find_matches C call vs Python post-processing)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth
from MTM import _lib
from MTM.distributed import merge_and_nms
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3_32"
img, units, plants = synth.make_config(cfg)
method, thr = (3, 0.9) if cfg == "cfg5" else (5, 0.5)
ctx = _lib.Context(0)
ctx.set_image(img)
ctx.set_templates([(u[1], u[2] if len(u) >= 3 else None) for u in units], method)
for it in range(4):
    t0 = time.perf_counter(); raw = ctx.find_matches(0, thr); t1 = time.perf_counter()
    tm = ctx.timing(); hits = merge_and_nms(raw.copy(), units, method, float("inf"), thr, 0.25); t2 = time.perf_counter()
    print("find_matches %.3f ms (gpu kernels %.3f) | post %.3f ms | raw hits %d -> %d" % ((t1 - t0) * 1e3, tm["total_ms"], (t2 - t1) * 1e3, len(raw), len(hits)), flush=True)
