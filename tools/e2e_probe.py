#!/usr/bin/env python3
"""Where one cold MTM.matchTemplates call spends its time (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth, MTM
from MTM import _lib
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3_32"
img, units, plants = synth.make_config(cfg)
ctx = _lib.default_context()
tl = [(u[1], u[2] if len(u) >= 3 else None) for u in units]
method, thr = (3, 0.9) if cfg == "cfg5" else (5, 0.5)
def t(f, n=8):
    f(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
print("%s: set_image %.3f | set_templates %.3f | find %.3f | matchTemplates %.3f ms" % (
    cfg, t(lambda: ctx.set_image(img)), t(lambda: ctx.set_templates(tl, method)), t(lambda: ctx.find_matches(0, thr)),
    t(lambda: MTM.matchTemplates(units, img, method=method, score_threshold=thr))), flush=True)
