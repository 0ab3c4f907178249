#!/usr/bin/env python3
"""Determinism hunt, sequence mode (GPU box): a run of fuzz cases through the DEFAULT context (as tools/fuzz_parity.py does:
one context, image type and template sets changing from call to call), passes repeated; every call's hit list is compared
with the same case's list of the first pass.  On a difference the case is run again at once (a state that persists, or a
race that does not?) and the differing templates are listed.
usage: flake_hunt2.py [first_seed [count [passes]]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
argv, sys.argv = sys.argv, sys.argv[:1]
import numpy as np
import fuzz_parity as F
import MTM
from MTM import _lib
first = int(argv[1]) if len(argv) > 1 else 7120
count = int(argv[2]) if len(argv) > 2 else 56
passes = int(argv[3]) if len(argv) > 3 else 30
pollute = os.environ.get("FLAKE_POLLUTE", "1") != "0"
if pollute:
    import torch


def junk():
    if not pollute:
        return
    a = torch.empty(1 << 27, dtype=torch.float32, device="cuda")
    a.uniform_(-1e38, 1e38)
    a[::7] = float("nan")
    torch.cuda.synchronize()
    del a
    torch.cuda.empty_cache()


def run(case):
    img, lt, method, thr, n_obj, box = case
    kw = dict(method=method, N_object=n_obj, searchBox=box)
    if thr is not None:
        kw["score_threshold"] = thr
    try:
        return [(h[0], tuple(h[1]), float(h[2])) for h in MTM.findMatches(lt, img, **kw)]
    except Exception as ex:  # noqa: BLE001
        return [("exception", (0, 0, 0, 0), hash(type(ex).__name__) % 1000)]


cases = [F.make_case(s) for s in range(first, first + count)]
ref = [None] * count
bad = 0
for ps in range(passes):
    if ps % 3 == 1:
        junk()
    order = range(count) if ps % 2 == 0 else list(range(count))[::-1] if ps % 4 == 1 else np.random.default_rng(ps).permutation(count)
    for i in order:
        got = run(cases[i])
        if ref[i] is None:
            ref[i] = got
            continue
        if got != ref[i]:
            bad += 1
            tm = _lib.default_context().timing()
            again = run(cases[i])
            g, e = set(got), set(ref[i])
            names = sorted({h[0] for h in g ^ e})
            img, lt, method, thr, n_obj, box = cases[i]
            shapes = {t[0]: (t[1].shape, len(t) > 2) for t in lt}
            print("pass %d seed %d (%s %s, method %d, thr %s, N %s, box %s): %d records vs %d, %d differ; templates %s; again -> %s; kernel_used %s hits_only %s f32_route %s" % (
                ps, first + i, img.shape, img.dtype, method, thr, n_obj, box, len(got), len(ref[i]), len(g ^ e),
                [(n, shapes.get(n)) for n in names][:6], "same as first pass" if again == ref[i] else "differs again" if again != got else "same wrong list",
                tm["kernel_used"], tm["hits_only"], tm["f32_route"]), flush=True)
            ex = sorted(g - e)[:3], sorted(e - g)[:3]
            print("      only now %s | only first %s" % ex, flush=True)
    print("pass %d done, %d differences so far" % (ps, bad), flush=True)
print("flake hunt (sequence): %d differing calls" % bad)
