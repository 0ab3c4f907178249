#!/usr/bin/env python3
"""Kernel time for few-template calls, row-multiplexed mode on/off (GPU box): MTM_ROW_MUX=0/1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth
from MTM import _lib
for hw, name in (((1080, 1920), "1080p"), ((2160, 3840), "4K")):
    for n in (1, 2, 4, 8, 16):
        img, units, plants = synth.make_workload(seed=5, image_hw=hw, n_base=n, templ=64, noisy_per_unit=1)
        ctx = _lib.Context(0)
        ctx.set_image(img); ctx.set_templates([(u[1], None) for u in units], 5)
        for i in range(60): ctx.find_matches(0, 0.5)
        ts = []
        for i in range(20):
            h = ctx.find_matches(0, 0.5); t = ctx.timing(); ts.append((t["ncc_kernel_ms"], t["total_ms"]))
        k, tot = np.median([a for a, b in ts]), np.median([b for a, b in ts])
        print("ROW_MUX=%s %s x %2d templates: ncc %.4f ms, gpu total %.4f ms, hits %d" % (os.environ.get("MTM_ROW_MUX", "1"), name, n, k, tot, len(h)), flush=True)
