#!/bin/bash
# kernel time under the runtime probes of ncc_mfma_kernel (results invalid while probing)
for envs in "X=0" "MTM_MFMA_DBG=2" "MTM_MFMA_DBG=10" "MTM_MFMA_DBG=6" "MTM_HITS_ONLY=0" "MTM_HITS_ONLY=0 MTM_MFMA_DBG=2"; do
  env $envs python - "$envs" <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "multitemplatematching-python_amd"))
import numpy as np, synth
from MTM import _lib
img, units, plants = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.Context(0)
ctx.set_image(img); ctx.set_templates([(u[1], None) for u in units], 5)
for i in range(150): ctx.find_matches(0, 0.5)
ts = []
for i in range(40):
    ctx.find_matches(0, 0.5); ts.append(ctx.timing()["ncc_kernel_ms"])
print("%-36s ncc median %.4f" % (sys.argv[1], float(np.median(ts))), flush=True)
PY
done
