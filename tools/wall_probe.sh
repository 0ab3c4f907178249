#!/bin/bash
# wall-clock and GPU time of mtm_find_matches per library variant (same box)
for r in 1 2; do for so in "$@"; do
  cp $so multitemplatematching-python_amd/MTM/libmtm_hip.so
  python - "$so" <<'PY'
import sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "multitemplatematching-python_amd"))
import numpy as np, synth
from MTM import _lib
img, units, plants = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.Context(0)
ctx.set_image(img); ctx.set_templates([(u[1], None) for u in units], 5)
for i in range(150): ctx.find_matches(0, 0.5)
w, g, k = [], [], []
for i in range(80):
    t0 = time.perf_counter(); ctx.find_matches(0, 0.5); w.append((time.perf_counter() - t0) * 1e3)
    t = ctx.timing(); g.append(t["total_ms"]); k.append(t["ncc_kernel_ms"])
print("%-30s wall %.4f  gpu total %.4f  ncc %.4f  (wall - ncc = %.1f us)" % (sys.argv[1], np.median(w), np.median(g), np.median(k), (np.median(w) - np.median(k)) * 1e3), flush=True)
PY
done; done
