#!/usr/bin/env python3
"""One named workload, a few calls - the command rocprofv3 wraps in tools/profile_workloads.sh (GPU box).
    workload.py <name> [calls]
names: cfg2 cfg3 cfg4 cfg5 (BASELINE configs; cfg4 = its 256 units on one GPU), u16_4k32 / f32_4k32 (4K x 32 templates 64x64 as uint16 / float32 pixels),
f64_1080p8 (float32 pixels on the float64 kernel, MTM_OPT_F32_MFMA = 0), slab_414 (2048^2 x one 414x400 template: the
reference's published benchmark shape, slabs on the MFMA kernel), dense_4k32 (photograph-like image: map mode + peak pass),
dense_4k32_nms (the same through mtm_find_matches_image_nms)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth
from MTM import _lib

name = sys.argv[1]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = _lib.Context(0)
method, thr = 5, 0.5
mode = _lib.PEAKS_LOCAL
if name in ("cfg2", "cfg3", "cfg4", "cfg5"):
    img, units, _ = synth.make_config(name)
    method, thr = (3, 0.9) if name == "cfg5" else (5, 0.5)
    tl = [(u[1], u[2] if len(u) >= 3 else None) for u in units]
elif name in ("u16_4k32", "f32_4k32", "f32_4k32_raw_n1", "f32_4k32_raw_n1_f64"):
    img, units, _ = synth.make_config("cfg3_32")
    dt = np.uint16 if name.startswith("u16") else np.float32
    img = img.astype(dt) * 257
    tl = [(u[1].astype(dt) * 257, None) for u in units]
    if "raw_n1" in name:           # TM_CCOEFF (raw sums), N_object == 1: the refined extremum on the bf16 matrix cores (round 4)
        method, mode = 4, _lib.PEAKS_GLOBAL
        if name.endswith("_f64"):  # ... against the float64 kernel
            ctx.set_option(_lib.OPT_F32_MFMA, 0)
elif name == "f64_1080p8":
    img, units, _ = synth.make_config("cfg2")
    img = img.astype(np.float32) * 0.5
    tl = [(u[1].astype(np.float32) * 0.5, None) for u in units]
    ctx.set_option(_lib.OPT_F32_MFMA, 0)
elif name == "slab_414":
    img = synth.smooth_u8(21, (2048, 2048))
    tl = [(np.ascontiguousarray(img[300:700, 500:914]), None)]
    thr = 0.9
elif name in ("dense_4k32", "dense_4k32_nms"):      # (_nms: search + non-maxima suppression in one native call)
    img = synth.smooth_u8(11, (2160, 3840))
    tl = [(u[1], None) for u in synth.cut_templates(5, img, 32, 64)]
    if name.endswith("_nms"):
        ctx.search = lambda tl_, img_, method_, mode_, thr_: ctx.search_nms(tl_, img_, method_, thr_, 0.25)
else:
    sys.exit("unknown workload " + name)
thr = float(os.environ.get("WL_THR", thr))       # (probing how a workload's time depends on its threshold)
ctx.search(tl, img, method, mode, thr)          # placement, allocation
for _ in range(3):
    ctx.search(tl, img, method, mode, thr)
st = []
for _ in range(calls):
    t0 = time.perf_counter()
    h = ctx.search(tl, img, method, mode, thr)
    st.append(time.perf_counter() - t0)
tm = ctx.timing()
px = img.shape[0] * img.shape[1]
macs = sum((img.shape[0] - t.shape[0] + 1) * (img.shape[1] - t.shape[1] + 1) * t.shape[0] * t.shape[1] * (1 if t.ndim == 2 else t.shape[2])
           for t, _ in tl)
algo = img.nbytes + sum(t.nbytes + (0 if m is None else m.nbytes) for t, m in tl)
import json
print(json.dumps({"workload": name, "image": list(img.shape), "dtype": str(img.dtype), "units": len(tl), "method": method,
                  "median_ms_per_call": round(float(np.median(st)) * 1e3, 4), "hits": int(len(h)),
                  "gpu_ms": round(float(tm["total_ms"]), 4), "ncc_kernel_ms": round(float(tm["ncc_kernel_ms"]), 4),
                  "ncc_launches": int(tm["ncc_launches"]), "kernel_used": int(tm["kernel_used"]), "hits_only": int(tm["hits_only"]),
                  "f32_route": int(tm["f32_route"]), "algorithmic_macs": int(macs), "input_bytes": int(algo),
                  "tmacs_in_score_kernels": round(macs / max(tm["ncc_kernel_ms"], 1e-9) / 1e9, 2)}))
