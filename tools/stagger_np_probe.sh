#!/bin/bash
# kernel time of the north-star launch with the first wave of blocks staggered
# (MTM_MFMA_STAGGER_NP = s_sleep(127) count of the second block on each CU); same box, same build
for r in 1 2; do for cfg in "0 0" "3 0" "6 0" "9 0" "6 2" "6 1"; do
  set -- $cfg
  MTM_MFMA_STAGGER_NP=$1 MTM_MFMA_STAGGER_MODE=$2 python - "$cfg" <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "multitemplatematching-python_amd"))
import numpy as np, synth
from MTM import _lib
img, units, plants = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.Context(0)
ctx.set_image(img); ctx.set_templates([(u[1], None) for u in units], 5)
for i in range(150): ctx.find_matches(0, 0.5)
ts = []
for i in range(60):
    ctx.find_matches(0, 0.5); ts.append(ctx.timing()["ncc_kernel_ms"])
print("stagger/mode %-8s ncc median %.4f min %.4f" % (sys.argv[1], float(np.median(ts)), min(ts)), flush=True)
PY
done; done
