#!/bin/bash
# A/B kernel time of library builds on the same GPU box: tools/ab_probe.sh a.so b.so [rounds]
A=$1; B=$2; N=${3:-3}
for r in $(seq 1 $N); do for so in $A $B; do
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
ts = []
for i in range(60):
    ctx.find_matches(0, 0.5); ts.append(ctx.timing()["ncc_kernel_ms"])
print("%-40s ncc median %.4f min %.4f" % (sys.argv[1], float(np.median(ts)), min(ts)), flush=True)
PY
done; done
