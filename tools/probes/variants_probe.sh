#!/bin/bash
# usage: tools/probes/variants_probe.sh <variant.so>[:ENV=VAL[,ENV=VAL...]] ...   (GPU box)
# steady-clock kernel time of the north-star launch for each library build (build/variants/*.so,
# made with MTM_EXTRA_FLAGS=-DMTM_PROBE_*); the scratch copy's in-tree library is overwritten
for spec in "$@"; do
  so="${spec%%:*}"; envs=""
  if [[ "$spec" == *:* ]]; then envs="${spec#*:}"; envs="${envs//,/ }"; fi
  cp "$so" multitemplatematching-python_amd/MTM/libmtm_hip.so
  env $envs python - "$spec" <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "multitemplatematching-python_amd"))
import numpy as np, synth
from MTM import _lib
img, units, plants = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.Context(0)
ctx.set_image(img); ctx.set_templates([(u[1], None) for u in units], 5)
for i in range(120): ctx.find_matches(0, 0.5)
ts = []
for i in range(40):
    ctx.find_matches(0, 0.5); ts.append(ctx.timing()["ncc_kernel_ms"])
print("%-50s ncc median %.4f min %.4f" % (sys.argv[1], float(np.median(ts)), min(ts)), flush=True)
PY
done
