#!/usr/bin/env python3
"""The reference's own tutorial workload at scale: the skimage coins photograph tiled to ~4K, the tutorial's two coin
templates x 4 rotations, threshold 0.5 (Tutorial1/2).  Per-call time, regime, hit counts - a real photograph, not
synthetic noise.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import MTM
from MTM import _lib
coins = np.load(os.path.join(ROOT, "tests", "golden", "coins.npz"))["image"]
small, big = coins[37:37 + 38, 80:80 + 41], coins[14:14 + 59, 302:302 + 65]
img = np.ascontiguousarray(np.tile(coins, (7, 10)))                 # 2121 x 3840
lt = [("%s_%d" % (n, 90 * k), np.ascontiguousarray(np.rot90(t, k))) for n, t in (("small", small), ("big", big)) for k in range(4)]
ctx = _lib.default_context()
for thr, n_obj in ((0.5, float("inf")), (0.7, float("inf")), (0.5, 24 * 70)):
    for _ in range(20):
        h = MTM.matchTemplates(lt, img, score_threshold=thr, maxOverlap=0.25, N_object=n_obj)
    ts, modes = [], []
    for _ in range(30):
        t = time.perf_counter(); h = MTM.matchTemplates(lt, img, score_threshold=thr, maxOverlap=0.25, N_object=n_obj); ts.append(time.perf_counter() - t)
        modes.append(ctx.timing()["hits_only"])
    tm = ctx.timing()
    print("coins x70 (%dx%d), %d templates, thr %.1f, N_object %s: call median %.3f ms, gpu %.3f ms (score %.3f, peaks %.3f), hits-only calls %d/30, raw peaks %d, hits %d" % (
        img.shape[1], img.shape[0], len(lt), thr, n_obj, np.median(ts) * 1e3, tm["total_ms"], tm["score_ms"], tm["peaks_ms"], sum(modes), tm["n_hits"], len(h)), flush=True)
