#!/usr/bin/env python3
"""Two size classes of 16 templates each (48x64 and 64x48: what rot90 of non-square crops gives) on a 4K image: per-call
time with the heaviest class running under the banded upload (default) against MTM_UPLOAD_BANDS=1.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth
from MTM import _lib
img = synth.rand_u8(31, 0, (2160, 3840))
rng = np.random.default_rng(5)
tl = []
for k in range(int(os.environ.get("PROBE_N", "16"))):
    y, x = int(rng.integers(0, 2000)), int(rng.integers(0, 3700))
    t = np.ascontiguousarray(img[y:y + 48, x:x + 64])
    tl += [(t, None), (np.ascontiguousarray(np.rot90(t)), None)]
ctx = _lib.Context(0)
for _ in range(30):
    h = ctx.search(tl, img, 5, _lib.PEAKS_LOCAL, 0.5)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); h = ctx.search(tl, img, 5, _lib.PEAKS_LOCAL, 0.5); ts.append(time.perf_counter() - t0)
tm = ctx.timing()
print("bands=%s lanes=%s: median %.4f ms, gpu %.3f ms, launches %d, hits %d" % (os.environ.get("MTM_UPLOAD_BANDS", "default"),
      os.environ.get("MTM_CLASS_LANES", "default"), np.median(ts) * 1e3, tm["total_ms"], tm["ncc_launches"], len(h)))
