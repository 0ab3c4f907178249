#!/usr/bin/env python3
"""Determinism hunt (GPU box): fuzz cases run again and again, each time on a FRESH context whose device buffers come out of
memory that was just filled with junk (huge floats, NaN patterns), records compared byte for byte with the first run's.
A kernel that reads what it never wrote, or a missing stream dependency, shows up as a difference.
usage: flake_hunt.py [first_seed [count [reps]]]    FLAKE_POLLUTE=0: no junk"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
argv, sys.argv = sys.argv, sys.argv[:1]
import numpy as np
import fuzz_parity as F
import MTM
from MTM import _lib
first = int(argv[1]) if len(argv) > 1 else 7160
count = int(argv[2]) if len(argv) > 2 else 10
reps = int(argv[3]) if len(argv) > 3 else 20
pollute = os.environ.get("FLAKE_POLLUTE", "1") != "0"
if pollute:
    import torch


def junk():
    if not pollute:
        return
    a = torch.empty(1 << 27, dtype=torch.float32, device="cuda")     # 512 MB
    a.uniform_(-1e38, 1e38)
    a[::7] = float("nan")
    b = torch.randint(0, 2 ** 31 - 1, (1 << 26,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    del a, b
    torch.cuda.empty_cache()


bad_total = 0
for seed in range(first, first + count):
    img, lt, method, thr, n_obj, box = F.make_case(seed)
    try:
        image, _, _ = MTM._validate_search(lt, img, n_obj, box)
    except Exception:
        continue
    args = (lt, image, method, n_obj, 0.5 if thr is None else thr)
    ref = None
    bad = 0
    for rep in range(reps):
        junk()
        ctx = _lib.Context(0)
        try:
            try:
                r = MTM._raw_matches(*args, context=ctx).copy()
            except Exception as ex:  # noqa: BLE001
                r = np.frombuffer(repr(ex).encode(), np.uint8)
            if ref is None:
                ref = r
            elif r.tobytes() != ref.tobytes():
                bad += 1
                if bad <= 2:
                    if len(r) == len(ref) and r.dtype == ref.dtype and r.dtype.names:
                        k = [i for i in range(len(r)) if r[i].tobytes() != ref[i].tobytes()]
                        print("   seed %d rep %d: %d of %d records differ, first %r vs %r" % (seed, rep, len(k), len(r), r[k[0]], ref[k[0]]), flush=True)
                    else:
                        print("   seed %d rep %d: %d records vs %d" % (seed, rep, len(r), len(ref)), flush=True)
        finally:
            ctx.close()
    bad_total += bad
    print("seed %d (%s %s, %d templates, method %d): %d of %d repetitions differ from the first" % (
        seed, img.shape, img.dtype, len(lt), method, bad, reps - 1), flush=True)
print("flake hunt: %d differing repetitions" % bad_total)
