#!/usr/bin/env python3
"""N matchTemplates calls on the bench workload (4K x 32 templates) - the command a rocprofv3 timeline wraps.
    loop_calls.py [pinned=0|1] [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth, MTM
pinned = len(sys.argv) > 1 and sys.argv[1] == "1"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 60
img, units, _ = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
if pinned:
    p = MTM.pinned_empty(img.shape, img.dtype)
    p[...] = img
    img = p
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    MTM.matchTemplates(units, img)
ts = []
for _ in range(calls):
    t = time.perf_counter(); MTM.matchTemplates(units, img); ts.append(time.perf_counter() - t)
print("pinned=%d bands=%s median %.4f ms min %.4f" % (pinned, os.environ.get("MTM_UPLOAD_BANDS", "default"), np.median(ts) * 1e3, min(ts) * 1e3))
