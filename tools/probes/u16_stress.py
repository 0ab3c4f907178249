#!/usr/bin/env python3
"""The uint16 two-class call of test_uint16_many_templates repeated U16_N times in one process; every raw record list is
compared with the first one (and the first one with the oracle).  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import MTM
from MTM import _lib
rng = np.random.default_rng(4242)
H, W = 150, 333
img = rng.integers(0, 65536, (H, W), dtype=np.uint16)
img[70:100, 40:120] = 777
lt = []
for i in range(37):
    y, x = int(rng.integers(0, H - 20)), int(rng.integers(0, W - 70))
    t = img[y:y + 20, x:x + 70].copy()
    if i % 3 == 0:
        t = np.clip(t.astype(np.int64) + rng.integers(-2000, 2000, t.shape), 0, 65535).astype(np.uint16)
    lt.append(("w%d" % i, t))
for i in range(18):
    y, x = int(rng.integers(0, H - 70)), int(rng.integers(0, W - 12))
    lt.append(("t%d" % i, img[y:y + 70, x:x + 12].copy()))
n = int(os.environ.get("U16_N", "2000"))
ctx = _lib.Context(0)
units = [(t, None) for _, t in lt]
ref = None
bad = 0
for rep in range(n):
    r = ctx.search(units, img, 5, _lib.PEAKS_LOCAL, 0.6).copy()
    if ref is None:
        ref = r
        continue
    if r.tobytes() != ref.tobytes():
        bad += 1
        if bad <= 3:
            a = set(map(tuple, r.tolist())); b = set(map(tuple, ref.tolist()))
            print("   rep %d: %d records vs %d; only now %s | only first %s" % (rep, len(r), len(ref), sorted(a - b)[:3], sorted(b - a)[:3]), flush=True)
print("%s: %d of %d calls differ from the first (%d records)" % (os.environ.get("U16_TAG", "default"), bad, n - 1, len(ref)), flush=True)
