#!/bin/bash
# time of everything but the ncc kernel (statistics + small kernels) per library variant
for so in "$@"; do
  cp $so multitemplatematching-python_amd/MTM/libmtm_hip.so
  python - "$so" <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "multitemplatematching-python_amd"))
import numpy as np, synth
from MTM import _lib
img, units, plants = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.Context(0)
ctx.set_image(img); ctx.set_templates([(u[1], None) for u in units], 5)
for i in range(150): ctx.find_matches(0, 0.5)
a, b = [], []
for i in range(60):
    ctx.find_matches(0, 0.5); t = ctx.timing(); a.append(t["score_ms"] - t["ncc_kernel_ms"]); b.append(t["total_ms"])
print("%-32s stats+gaps %.4f ms, gpu total %.4f ms" % (sys.argv[1], float(np.median(a)), float(np.median(b))), flush=True)
PY
done
