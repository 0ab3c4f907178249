#!/usr/bin/env python3
"""The scenario of tests/test_gpu_parity.py::test_uint16_many_templates, hit list against the oracle, with the differing
hits and their map values printed (GPU box).  U16_REPS: repetitions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import MTM
from MTM import _lib
import mtm_oracle as O
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
f32img = img.astype(np.float32)
exp = O.find_matches([(n, t.astype(np.float32)) for n, t in lt], f32img, method=5, score_threshold=0.6)
e = {(h[0], tuple(h[1])): float(h[2]) for h in exp}
tag = os.environ.get("U16_TAG", "default")
for rep in range(int(os.environ.get("U16_REPS", "3"))):
    got = MTM.findMatches(lt, img, method=5, score_threshold=0.6)
    g = {(h[0], tuple(h[1])): float(h[2]) for h in got}
    tm = _lib.default_context().timing()
    only_g, only_e = sorted(g.keys() - e.keys()), sorted(e.keys() - g.keys())
    bad = [(k, g[k], e[k]) for k in g.keys() & e.keys() if abs(g[k] - e[k]) > 1e-6]
    print("%s rep %d: got %d exp %d | only got %s | only exp %s | score diffs %s | kernel_used %s hits_only %s" % (
        tag, rep, len(g), len(e), [(k, g[k]) for k in only_g][:4], [(k, e[k]) for k in only_e][:4], bad[:3], tm["kernel_used"], tm["hits_only"]), flush=True)
    for k in only_g[:2]:
        name, (x, y, w, h) = k
        idx = [n for n, _ in lt].index(name)
        m = O.match_template(f32img, lt[idx][1].astype(np.float32), 5)
        ys, xs = slice(max(0, y - 1), y + 2), slice(max(0, x - 1), x + 2)
        print("   oracle map around it:\n", np.array2string(m[ys, xs], precision=7))
        c = _lib.Context(0)
        c.set_image(img); c.set_templates([(t, None) for _, t in lt], 5)
        gm = c.score_map(idx, m.shape)
        print("   product map (one-template launch) around it:\n", np.array2string(gm[ys, xs], precision=7))
        c.set_option(_lib.OPT_HITS_ONLY, 0)
        c.find_matches(_lib.PEAKS_LOCAL, 0.6)
        lm = c.last_score_map(idx, m.shape)
        print("   product map (whole-set launch) around it:\n", np.array2string(lm[ys, xs], precision=7), "max |diff| whole map", float(np.nanmax(np.abs(lm - m))))
        c.close()
