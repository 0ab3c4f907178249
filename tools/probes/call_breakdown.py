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
# fresh template bytes in every call: where the extra time goes (marshalling, mtm_set_templates, the find call with
# the placement inside it)
variants = []
for i in range(4):
    v = []
    for t, m in tl:
        t2 = t.copy(); t2[0, 0] ^= (i + 1); v.append((t2, m))
    variants.append(v)
for i in range(8):
    ctx.search(variants[i % 4], img, 5, _lib.PEAKS_LOCAL, 0.5)
rec_t, set_t, find_t = [], [], []
for i in range(80):
    v = variants[i % 4]
    t = time.perf_counter(); rec = ctx._records(v); t1 = time.perf_counter()
    _lib.check(ctx._lib.mtm_set_templates(ctx._h, rec.ctypes.data, len(v), 5), "mtm_set_templates"); t2 = time.perf_counter()
    ctx.find_matches_image(img, _lib.PEAKS_LOCAL, 0.5); t3 = time.perf_counter()
    rec_t.append(t1 - t); set_t.append(t2 - t1); find_t.append(t3 - t2)
same = []
for i in range(40):
    t = time.perf_counter(); ctx.search(variants[3], img, 5, _lib.PEAKS_LOCAL, 0.5); same.append(time.perf_counter() - t)
print("fresh templates | records %.3f ms + mtm_set_templates %.3f + find (placement inside) %.3f = %.3f | unchanged set: search %.3f" % (
    med(rec_t), med(set_t), med(find_t), med(rec_t) + med(set_t) + med(find_t), med(same)))
