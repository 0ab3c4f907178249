#!/usr/bin/env python3
"""One case of tools/fuzz_parity.py by seed: product against oracle, hit by hit (GPU box).  FUZZ_ONLY=t27: that template alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.argv, argv = sys.argv[:1], sys.argv
import fuzz_parity as F
import numpy as np
import MTM
import mtm_oracle as O
seed = int(argv[1])
img, lt, method, thr, n_obj, box = F.make_case(seed)
only = os.environ.get("FUZZ_ONLY")
if only:
    lt = [u for u in lt if u[0] in only.split(",")]
kw = dict(method=method, N_object=n_obj, searchBox=box)
if thr is not None:
    kw["score_threshold"] = thr
got = MTM.findMatches(lt, img, **kw)
oi, ol = F.as_oracle(img, lt)
exp = O.find_matches(ol, oi, **kw)
g = {(h[0], tuple(h[1])): float(h[2]) for h in got}
e = {(h[0], tuple(h[1])): float(h[2]) for h in exp}
bad = [(k, g[k], e[k]) for k in g.keys() & e.keys() if abs(g[k] - e[k]) > 2e-5 * max(1.0, abs(e[k]))]
from MTM import _lib
tm = _lib.default_context().timing()
print("seed %d %s: got %d exp %d, common %d, score differences %d, only-got %d, only-exp %d | kernel_used %s hits_only %s" % (
    seed, os.environ.get("FUZZ_TAG", ""), len(g), len(e), len(g.keys() & e.keys()), len(bad), len(g.keys() - e.keys()), len(e.keys() - g.keys()),
    tm["kernel_used"], tm["hits_only"]))
for b in bad[:6]:
    print("   ", b)
