#!/usr/bin/env python3
"""4K RGB image x 32 RGB templates 64x64, maps in memory (computeScoreMap's route) and hits-only: per call / score kernel.
    rgb_maps_probe.py [calls]        (MTM_LIB_PATH selects the library: same-box A/B of two builds)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd")); sys.path.insert(0, ROOT)
import numpy as np
import synth
from MTM import _lib
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g, units, _ = synth.make_config("cfg3_32")
img = np.ascontiguousarray(np.stack([g, np.roll(g, 7, axis=1), 255 - np.roll(g, 3, axis=0)], axis=2))
rng = np.random.default_rng(3)
tl = []
for k in range(32):
    y, x = int(rng.integers(0, img.shape[0] - 64)), int(rng.integers(0, img.shape[1] - 64))
    tl.append((np.ascontiguousarray(img[y:y + 64, x:x + 64]), None))
out = []
for method in (5, 3):
    for honly in (0, 1):
        ctx = _lib.Context(0)
        ctx.set_option(_lib.OPT_HITS_ONLY, honly)
        for _ in range(3):
            h = ctx.search(tl, img, method, _lib.PEAKS_LOCAL, 0.7)
        st = []
        for _ in range(calls):
            t0 = time.perf_counter(); h = ctx.search(tl, img, method, _lib.PEAKS_LOCAL, 0.7); st.append(time.perf_counter() - t0)
        tm = ctx.timing()
        out.append("method %d hits_only %d: %.3f ms per call, score kernel %.3f ms, %d hits, kernel_used %d, digest %08x" % (
            method, honly, float(np.median(st)) * 1e3, tm["ncc_kernel_ms"], len(h), tm["kernel_used"],
            int(np.bitwise_xor.reduce(h["score"].view(np.uint32).astype(np.uint64) * (h["x"].astype(np.uint64) + 1)) & 0xffffffff)))
        del ctx
print(" | ".join(out))
