#!/usr/bin/env python3
"""The order the process-per-GPU bench creates things in: context, one-rank RCCL communicator, THEN the first search."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth, MTM
from MTM import _lib
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
img, units, _ = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.default_context()
ctx.comm_init(_lib.comm_unique_id(), 1, 0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    MTM.matchTemplates(units, img)
ts, ks, gs = [], [], []
for _ in range(calls):
    t = time.perf_counter()
    raw = MTM._raw_matches(units, img, 5, float("inf"), 0.5, context=ctx).copy()
    allh = ctx.allgather_hits(raw)[0]
    ts.append(time.perf_counter() - t)
    tm = ctx.timing(); ks.append(tm["ncc_kernel_ms"]); gs.append(tm["total_ms"])
print("communicator first, search + all-gather of %d hits: median %.4f ms | ncc %.4f gpu %.4f (eager copy stream: %s)" % (
    len(allh), np.median(ts) * 1e3, np.median(ks), np.median(gs), os.environ.get("MTM_EAGER_COPY_STREAM", "1")), flush=True)
