#!/usr/bin/env python3
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
import synth, MTM
from MTM import _lib
img, units, plants = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.Context(0)
m = MTM.TemplateMatcher(units, score_threshold=0.5, maxOverlap=0.25, context=ctx)
frames = [np.ascontiguousarray(np.roll(img, 64 * k, axis=1)) for k in range(4)] * 4
list(m.match_stream(frames[:3]))
ts = []
t = time.perf_counter()
for h in m.match_stream(frames):
    t2 = time.perf_counter(); ts.append((t2 - t) * 1e3); t = t2
print("per image ms:", " ".join("%.2f" % x for x in ts), flush=True)
