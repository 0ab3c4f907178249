#!/usr/bin/env python3
"""Does a live RCCL communicator in the process change what the search itself costs?  The bench workload on a plain context,
before and after mtm_comm_init of a one-rank communicator on that context (never used by the calls).
    rccl_probe.py [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth, MTM
from MTM import _lib
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
img, units, _ = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.default_context()


def run(tag):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        MTM.matchTemplates(units, img)
    ts, ks, gs = [], [], []
    for _ in range(calls):
        t = time.perf_counter(); MTM.matchTemplates(units, img); ts.append(time.perf_counter() - t)
        tm = ctx.timing(); ks.append(tm["ncc_kernel_ms"]); gs.append(tm["total_ms"])
    print("%s: median %.4f ms | ncc %.4f gpu %.4f" % (tag, np.median(ts) * 1e3, np.median(ks), np.median(gs)), flush=True)


run("plain context")
ctx.comm_init(_lib.comm_unique_id(), 1, 0)
run("same context, one-rank RCCL communicator alive")
run("again")
