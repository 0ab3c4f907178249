#!/usr/bin/env python3
"""N_object == 1 (cv2.minMaxLoc): fused extremum (MTM_OPT_HITS_ONLY = 1) against score maps + extremum_kernel
(= 0) on the bench workload, on BASELINE config 2 (1080p x 8, row-multiplexed) and on RGB; GPU time per call
from the library's events and wall time.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "multitemplatematching-python_amd"))
import numpy as np, synth
from MTM import _lib
cases = (("4K x 32 grey", (2160, 3840), 32, 1), ("1080p x 8 grey", (1080, 1920), 8, 1),
         ("4K x 32 RGB", (2160, 3840), 32, 3), ("4K x 8 RGB", (2160, 3840), 8, 3))
for name, hw, n, ch in cases:
    img, units, plants = synth.make_workload(seed=3, image_hw=hw, n_base=n, templ=64, noisy_per_unit=3, channels=ch)
    ctx = _lib.Context(0)
    ctx.set_image(img)
    ctx.set_templates([(u[1], None) for u in units], 5)
    for i in range(60): ctx.find_matches(1, 0.5)
    for honly in (1, 0):
        ctx.set_option(_lib.OPT_HITS_ONLY, honly)
        for i in range(10): r = ctx.find_matches(1, 0.5)
        g = []
        t0 = time.perf_counter()
        for i in range(30):
            r = ctx.find_matches(1, 0.5); g.append(ctx.timing()["total_ms"])
        wall = (time.perf_counter() - t0) / 30 * 1e3
        print("%-16s fused=%d: gpu %.3f ms, wall %.3f ms, hits %d, best %.4f" %
              (name, honly, np.median(g), wall, len(r), float(r["score"].max())), flush=True)
    del ctx
