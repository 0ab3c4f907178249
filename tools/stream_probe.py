#!/usr/bin/env python3
"""Per-image wall-clock of MTM.TemplateMatcher: match() loop vs match_stream() (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth, MTM
from MTM import _lib
img, units, plants = synth.make_config("cfg3_32")
ctx = _lib.Context(0)
m = MTM.TemplateMatcher(units, score_threshold=0.5, context=ctx)
frames = [np.ascontiguousarray(np.roll(img, 64 * k, axis=1)) for k in range(4)] * 6
m.match(frames[0])
for rep in range(2):
    t0 = time.perf_counter(); a = [m.match(f) for f in frames]; t1 = time.perf_counter()
    b = list(m.match_stream(frames)); t2 = time.perf_counter()
    assert a == b
    t3 = time.perf_counter()
    for f in frames: ctx.set_image(f)
    t4 = time.perf_counter()
    n = len(frames)
    print("match loop %.3f ms/img | match_stream %.3f ms/img | set_image alone %.3f ms" % ((t1 - t0) / n * 1e3, (t2 - t1) / n * 1e3, (t4 - t3) / n * 1e3), flush=True)
