#!/usr/bin/env python3
"""N_object == 1 (cv2.minMaxLoc) on the bench workload: fused extremum (MTM_OPT_HITS_ONLY = 1) against score
maps + extremum_kernel (= 0); GPU time per call from the library's events and wall time.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multitemplatematching-python_amd"))
import numpy as np, synth
from MTM import _lib
img, units, plants = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.Context(0)
ctx.set_image(img)
for method in (5,):       # (SQDIFF_NORMED at threshold 0.5 floods the local-peaks warm-up below: not timed here)
    ctx.set_templates([(u[1], None) for u in units], method)
    for i in range(120): ctx.find_matches(0, 0.5)
    for honly in (1, 0, 1, 0):
        ctx.set_option(_lib.OPT_HITS_ONLY, honly)
        for i in range(10): r = ctx.find_matches(1, 0.5)
        g, k = [], []
        t0 = time.perf_counter()
        for i in range(50):
            r = ctx.find_matches(1, 0.5); t = ctx.timing(); g.append(t["total_ms"]); k.append(t["ncc_kernel_ms"])
        wall = (time.perf_counter() - t0) / 50 * 1e3
        print("method %d fused=%d: gpu %.3f ms (ncc %.3f), wall %.3f ms, hits %d, best %.4f" %
              (method, honly, np.median(g), np.median(k), wall, len(r), float(r["score"].max())), flush=True)
    ctx.set_option(_lib.OPT_HITS_ONLY, 1)
    for i in range(30): ctx.find_matches(0, 0.5)
    ts = []
    for i in range(60):
        ctx.find_matches(0, 0.5); ts.append(ctx.timing()["ncc_kernel_ms"])
    print("method %d local-peaks mode ncc median %.4f" % (method, float(np.median(ts))), flush=True)
