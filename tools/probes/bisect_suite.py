#!/usr/bin/env python3
"""Which earlier GPU test leaves the state that makes TARGET fail?  Binary search on the length of the prefix of the suite
(in collection order) that is run ahead of it, then the culprit alone + the target.  GPU box.
usage: bisect_suite.py <target node id substring>"""
import subprocess, sys
target_key = sys.argv[1]
out = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q"], capture_output=True, text=True).stdout
ids = [l.strip() for l in out.splitlines() if "::" in l]
ti = next(i for i, s in enumerate(ids) if target_key in s)
target, before = ids[ti], ids[:ti]
print("target", target, "preceded by", len(before), "tests", flush=True)


def fails(subset):
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + subset + [target], capture_output=True, text=True)
    tail = [l for l in r.stdout.splitlines() if "passed" in l or "failed" in l]
    bad = target.split("::")[-1] in "".join(l for l in r.stdout.splitlines() if l.startswith("FAILED"))
    print("   %d tests ahead -> %s %s" % (len(subset), "FAILS" if bad else "passes", tail[-1:] ), flush=True)
    return bad


if not fails(before):
    print("does not fail with the whole prefix")
    sys.exit(0)
lo, hi = 0, len(before)          # fails with before[:hi], assume passes with before[:lo]
while hi - lo > 1:
    mid = (lo + hi) // 2
    if fails(before[:mid]):
        hi = mid
    else:
        lo = mid
culprit = before[hi - 1]
print("culprit (last test of the shortest failing prefix):", culprit, flush=True)
print("culprit alone ahead of the target:", "FAILS" if fails([culprit]) else "passes")
