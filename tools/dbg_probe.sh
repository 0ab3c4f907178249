#!/bin/bash
# kernel time with and without the epilogue (MTM_MFMA_DBG=2: results invalid), hits-only and map mode;
# the other probes are compile-time: MTM_EXTRA_FLAGS=-DMTM_PROBE_{NO_A,NO_Q,NO_STAGE,NO_MFMA,FROZEN_A,NO_STORE,CHEAP_EPI}
for envs in "X=0" "MTM_MFMA_DBG=2" "MTM_HITS_ONLY=0" "MTM_HITS_ONLY=0 MTM_MFMA_DBG=2"; do
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
