#!/usr/bin/env python3
"""Sweep the register-blocking variants of ncc_dot4_kernel (and the other kernels) on one workload
and print kernel times.  Run on the GPU box:  python tools/tune_dot4.py [cfg] > gpurun_out/tune.log"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np  # noqa: E402
import synth  # noqa: E402
from MTM import _lib  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3_32"
img, units, plants = synth.make_config(cfg)
ctx = _lib.Context(0)
ctx.set_image(img)
ctx.set_templates([(u[1], None) for u in units], 5)
H, W = img.shape[:2]
macs = sum((H - u[1].shape[0] + 1) * (W - u[1].shape[1] + 1) * u[1].shape[0] * u[1].shape[1] for u in units)
res = []
for kernel, variants in ((2, [0, 2]), (3, [0])):
    ctx.set_option(_lib.OPT_KERNEL, kernel)
    for v in variants:
        if kernel == 2:
            ctx.set_option(_lib.OPT_DOT4_VARIANT, v)
        ts = []
        for it in range(4):
            hits = ctx.find_matches(0, 0.5)
            ts.append(ctx.timing())
        t = ts[-1]
        best = min(x["ncc_kernel_ms"] for x in ts[1:])
        row = dict(cfg=cfg, kernel=kernel, variant=v, ncc_ms=round(best, 3), score_ms=round(t["score_ms"], 3),
                   peaks_ms=round(t["peaks_ms"], 3), total_ms=round(t["total_ms"], 3), hits=len(hits),
                   tmacs=round(macs / best / 1e9, 1), used=t["kernel_used"])
        res.append(row)
        print(json.dumps(row), flush=True)
