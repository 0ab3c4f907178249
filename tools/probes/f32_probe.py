#!/usr/bin/env python3
"""Timing of the uint8, uint16 (byte-plane MFMA) and float32 (bf16-piece MFMA kernel; MTM_F32_MFMA=0: float64 kernel)
paths at 1080p x 8 and 4K x 32 (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth
from MTM import _lib
for cfg in ("cfg2", "cfg3_32"):
    img, units, plants = synth.make_config(cfg)
    ctx = _lib.Context(0)
    for name, im, tl in (("uint8", img, [(u[1], None) for u in units]),
                         ("uint16", img.astype(np.uint16) * 257, [(u[1].astype(np.uint16) * 257, None) for u in units]),
                         ("float32", img.astype(np.float32) * 257, [(u[1].astype(np.float32) * 257, None) for u in units])):
        ctx.set_image(im); ctx.set_templates(tl, 5)
        ctx.find_matches(0, 0.5)
        t0 = time.perf_counter(); h = ctx.find_matches(0, 0.5); dt = (time.perf_counter() - t0) * 1e3
        tm = ctx.timing()
        print("%s %s: find %.2f ms (gpu %.2f, ncc %.2f, kernel_used %d) hits %d" % (cfg, name, dt, tm["total_ms"], tm["ncc_kernel_ms"], tm["kernel_used"], len(h)), flush=True)
