#!/usr/bin/env python3
"""Decompose the MFMA kernel's time with the MTM_MFMA_DBG probes (results invalid while probing)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, json
sys.path.insert(0, os.path.join(%r, "multitemplatematching-python_amd"))
import synth
from MTM import _lib
img, units, plants = synth.make_config("cfg3_32")
ctx = _lib.Context(0)
ctx.set_image(img); ctx.set_templates([(u[1], None) for u in units], 5)
best = 1e9
for i in range(5):
    ctx.find_matches(0, 0.5); t = ctx.timing(); best = min(best, t["ncc_kernel_ms"])
print(json.dumps(dict(dbg=int(os.environ.get("MTM_MFMA_DBG", "0")), ncc_ms=round(best, 3), peaks_ms=round(t["peaks_ms"], 3), score_ms=round(t["score_ms"], 3))))
''' % ROOT
runs = [dict()]
for mode in ("0", "1", "2"):
    for st in ("50", "200"):
        runs.append(dict(MTM_MFMA_STAGGER_MODE=mode, MTM_MFMA_STAGGER=st))
for extra in runs:
    env = dict(os.environ, MTM_KERNEL="mfma", **extra)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(json.dumps(extra), r.stdout.strip() or r.stderr[-300:], flush=True)
