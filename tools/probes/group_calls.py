#!/usr/bin/env python3
"""N matchTemplates calls on the bench workload through a device GROUP of one device (mtm_group: worker thread, shard,
exchange, NMS inside the native call) - what BENCH_GROUP_SINGLE=1 bench.py times; the command a rocprofv3 timeline wraps.
    group_calls.py [rccl=0|1] [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
os.environ["MTM_DEVICES_FORCE_GROUP"] = "1"
import numpy as np
import synth, MTM
from MTM import _lib
rccl = len(sys.argv) > 1 and sys.argv[1] in ("1", "2")
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 60
img, units, _ = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
g = _lib.engine_for([0])
ranks = g.comm_init(strict=False) if rccl else 0
if len(sys.argv) > 1 and sys.argv[1] == "2":        # communicators alive, but the lists are merged on the host
    g.set_exchange("host")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    MTM.matchTemplates(units, img, devices=[0])
ts = []
for _ in range(calls):
    t = time.perf_counter(); hits = MTM.matchTemplates(units, img, devices=[0]); ts.append(time.perf_counter() - t)
tm = g.timing(0)
print("group of 1, exchange %s (ranks %d): median %.4f ms min %.4f | ncc %.4f gpu %.4f | %d hits" % (
    g.exchange_used(), ranks, np.median(ts) * 1e3, min(ts) * 1e3, tm["ncc_kernel_ms"], tm["total_ms"], len(hits)))
