#!/usr/bin/env python3
"""The reference's own published timings (tutorials/Benchmark.ipynb:234,383: 264 ms for one 414 x 400 template,
381 ms for three rotations of it, unknown Windows PC; tutorials/Tutorial3-SpeedingUp.ipynb:223: 125 ms for one
196 x 184 template on 2048 x 2048, N_object = 1) on a synthetic image of the same sizes: one MTM.matchTemplates
call, numpy in -> hits out, median of 20.  MTM_KERNEL=dot4 gives the VALU fallback the large templates took before
the slab decomposition.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth, MTM
img = synth.rand_u8(77, 0, (2048, 2048))
big = np.ascontiguousarray(img[700:1114, 900:1300])
for k, (y, x) in enumerate(((100, 100), (1500, 300), (300, 1500))):       # three more noisy copies
    noise = (synth.rand_u8(77, 10 + k, big.shape).astype(np.int32) % 81) - 40
    img[y:y + 414, x:x + 400] = np.clip(big.astype(np.int32) + noise, 0, 255)
mid = np.ascontiguousarray(img[300:496, 1000:1184])                       # 196 x 184
def med(fn, n=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3, r
k = os.environ.get("MTM_KERNEL", "auto")
t1, r1 = med(lambda: MTM.matchTemplates([("well", big)], img, N_object=4, score_threshold=0.4, maxOverlap=0.3, method=5))
rots = [("0", big), ("90", np.ascontiguousarray(np.rot90(big))), ("180", np.ascontiguousarray(np.rot90(big, 2)))]
t3, r3 = med(lambda: MTM.matchTemplates(rots, img, N_object=4, score_threshold=0.4, maxOverlap=0.3, method=5))
t0, r0 = med(lambda: MTM.matchTemplates([("t", mid)], img, N_object=1))
print("kernel=%s | 2048^2 x one 414x400: %.2f ms (%d hits; reference 264 ms) | x three rotations: %.2f ms (%d hits; reference 381 ms) | "
      "x one 196x184, N_object=1: %.2f ms (reference 125 ms)" % (k, t1, len(r1), t3, len(r3), t0))
