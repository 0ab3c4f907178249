#!/usr/bin/env python3
"""Where one MTM.matchTemplates call (numpy in -> hits out) spends its wall-clock on the bench workload:
Python host layer / native call / GPU kernels (HIP events).  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth, MTM
from MTM import _lib
img, units, _ = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.default_context()
tl = [(u[1], None) for u in units]
native = []
orig = ctx.search
def timed(*a):
    t = time.perf_counter(); r = orig(*a); native.append(time.perf_counter() - t); return r
ctx.search = timed
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.6:
    MTM.matchTemplates(units, img)
native.clear(); tot = []; gpu = []; ker = []
for _ in range(60):
    t = time.perf_counter(); MTM.matchTemplates(units, img); tot.append(time.perf_counter() - t)
    tm = ctx.timing(); gpu.append(tm["total_ms"]); ker.append(tm["ncc_kernel_ms"])
ctx.search = orig
st = []
for _ in range(60):
    t = time.perf_counter(); ctx.set_templates(tl, 5); st.append(time.perf_counter() - t)
up = []
for _ in range(20):
    t = time.perf_counter(); ctx.set_image(img); up.append(time.perf_counter() - t)
med = lambda x: float(np.median(x)) * 1e3
print("bands=%s prio=%s | call %.3f ms = python %.3f + native %.3f (set_templates alone %.3f) | gpu first-kernel->done %.3f, score kernels %.3f | set_image alone %.3f" % (
    os.environ.get("MTM_UPLOAD_BANDS", "default"), os.environ.get("MTM_COPY_PRIO", "1"), med(tot), med(tot) - med(native), med(native), med(st),
    float(np.median(gpu)), float(np.median(ker)), med(up)))
