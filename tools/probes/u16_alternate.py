#!/usr/bin/env python3
"""Two uint16 scenarios of the same shapes but different pixels (and, ALT_GEOM=1, a third of another geometry) alternate on
ONE context: whatever a call reads that it has not written itself holds the OTHER scenario's data.  Every call's records
are compared with the scenario's first result (itself checked against the oracle).  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from MTM import _lib
import mtm_oracle as O


def scenario(seed, H=150, W=333, dtype=np.uint16):
    rng = np.random.default_rng(seed)
    hi = 65536 if dtype == np.uint16 else 256
    img = rng.integers(0, hi, (H, W)).astype(dtype)
    img[70:100, 40:120] = 77
    lt = []
    for i in range(37):
        y, x = int(rng.integers(0, H - 20)), int(rng.integers(0, W - 70))
        t = img[y:y + 20, x:x + 70].copy()
        if i % 3 == 0:
            t = np.clip(t.astype(np.int64) + rng.integers(-hi // 32, hi // 32, t.shape), 0, hi - 1).astype(dtype)
        lt.append(t)
    for i in range(18):
        y, x = int(rng.integers(0, H - 70)), int(rng.integers(0, W - 12))
        lt.append(img[y:y + 70, x:x + 12].copy())
    return img, [(t, None) for t in lt]


dtype = np.uint8 if os.environ.get("ALT_DTYPE") == "uint8" else np.uint16
scen = [scenario(4242, dtype=dtype), scenario(99, dtype=dtype)]
if os.environ.get("ALT_GEOM"):
    scen.append(scenario(7, H=171, W=290, dtype=dtype))
n = int(os.environ.get("ALT_N", "1500"))
method, thr = 5, 0.6
ctx = _lib.Context(0)
refs = []
for img, units in scen:
    r = ctx.search(units, img, method, _lib.PEAKS_LOCAL, thr).copy()
    exp = O.find_matches([("t%d" % i, u[0].astype(np.float32)) for i, u in enumerate(units)], img.astype(np.float32), method=method, score_threshold=thr)
    e = sorted((int(h[0][1:]), h[1][0], h[1][1]) for h in exp)
    g = sorted((int(a["templ_idx"]), int(a["x"]), int(a["y"])) for a in r)
    if e != g:
        print("   first result of a scenario differs from the oracle: %d vs %d records, e.g. %s" % (len(g), len(e), sorted(set(g) ^ set(e))[:3]))
    refs.append(r)
bad = 0
for rep in range(n):
    k = rep % len(scen)
    img, units = scen[k]
    r = ctx.search(units, img, method, _lib.PEAKS_LOCAL, thr)
    if r.tobytes() != refs[k].tobytes():
        bad += 1
        if bad <= 3:
            a = set(map(tuple, r.tolist())); b = set(map(tuple, refs[k].tolist()))
            print("   call %d (scenario %d): %d records vs %d; only now %s | only reference %s" % (rep, k, len(r), len(refs[k]), sorted(a - b)[:3], sorted(b - a)[:3]), flush=True)
print("%s: %d of %d alternating calls differ from their scenario's reference" % (os.environ.get("ALT_TAG", "default"), bad, n), flush=True)
