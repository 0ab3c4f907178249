#!/usr/bin/env python3
"""One step of the process-per-GPU form at ONE rank (a one-GPU box): context, one-rank communicator, then per step either
the four steps from Python (search, renumber, all-gather, merge + NMS + hit list) or the one native call
(mtm_find_matches_image_sharded_nms) + hit list.   sharded_step_probe.py [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth, MTM
from MTM import _lib
from MTM.distributed import HitExchange, merge_and_nms, _u8_units
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
img, units, _ = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.default_context()
ctx.comm_init(_lib.comm_unique_id(), 1, 0)
gidx = np.arange(len(units), dtype=np.int32)
inf = float("inf")
sub_units = _u8_units(units, img, 5)


def old():
    raw = MTM._raw_matches(units, img, 5, inf, 0.5, context=ctx).copy()
    raw["templ_idx"] = gidx[raw["templ_idx"]]
    return merge_and_nms(ctx.allgather_hits(raw)[0], units, 5, inf, 0.5, 0.25)


def new():
    return MTM._to_hit_list(ctx.search_sharded_nms(sub_units, img, 5, 0.5, 0.25, -1, gidx), units, 0, 0)


assert old() == new()
for name, fn in (("four steps from Python", old), ("one native call", new), ("four steps from Python", old), ("one native call", new)):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        fn()
    ts = []
    for _ in range(calls):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    print("sharded step at one rank, %-24s median %.4f ms min %.4f" % (name + ":", np.median(ts) * 1e3, min(ts) * 1e3), flush=True)
