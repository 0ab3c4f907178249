#!/bin/bash
for envs in "X=0" "MTM_MFMA_PERSISTENT=1" "MTM_MFMA_PERSISTENT=1 MTM_MFMA_STAGGER=0" "MTM_MFMA_PERSISTENT=1 MTM_MFMA_STAGGER=2" "MTM_MFMA_PERSISTENT=1 MTM_MFMA_STAGGER=4" "MTM_MFMA_PERSISTENT=1 MTM_MFMA_STAGGER=4 MTM_MFMA_STAGGER_MODE=2" "MTM_MFMA_PERSISTENT=1 MTM_MFMA_STAGGER=8"; do
  echo "== $envs"
  env $envs python - <<'PY'
import sys, os, json
sys.path.insert(0, os.path.join(os.getcwd(), "multitemplatematching-python_amd"))
import numpy as np, synth
from MTM import _lib
img, units, plants = synth.make_config("cfg3_32")
ctx = _lib.Context(0)
ctx.set_image(img); ctx.set_templates([(u[1], None) for u in units], 5)
for i in range(100): ctx.find_matches(0, 0.5)
ts = []
for i in range(30):
    ctx.find_matches(0, 0.5); ts.append(ctx.timing()["ncc_kernel_ms"])
print("ncc median %.4f min %.4f" % (float(np.median(ts)), min(ts)))
PY
done
