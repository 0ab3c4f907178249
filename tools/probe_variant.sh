#!/bin/bash
# usage: tools/probe_variant.sh <variant.so>[:ENV=VAL[,ENV=VAL...]] ...
# (GPU box; overwrites the in-tree library of the scratch copy)
for spec in "$@"; do
  so="${spec%%:*}"; envs=""
  if [[ "$spec" == *:* ]]; then envs="${spec#*:}"; envs="${envs//,/ }"; fi
  cp "$so" multitemplatematching-python_amd/MTM/libmtm_hip.so
  echo "== $spec"
  env MTM_KERNEL=mfma $envs python - <<'PY'
import sys, os, json
sys.path.insert(0, os.path.join(os.getcwd(), "multitemplatematching-python_amd"))
import synth
from MTM import _lib
img, units, plants = synth.make_config(os.environ.get("PROBE_CFG", "cfg3_32"))
ctx = _lib.Context(0)
ctx.set_image(img); ctx.set_templates([(u[1], None) for u in units], 5)
best = 1e9; tot = 1e9
for i in range(8):
    ctx.find_matches(0, 0.5); t = ctx.timing(); best = min(best, t["ncc_kernel_ms"]); tot = min(tot, t["total_ms"])
print(json.dumps(dict(ncc_ms=round(best, 4), total_ms=round(tot, 4), peaks_ms=round(t["peaks_ms"], 4), hits=int(t["n_hits"]))))
PY
done
