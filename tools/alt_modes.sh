#!/bin/bash
# GPU parity suite under every alternative compute route of the library (GPU box).  Round 6: 20 MTM_* variables are left in
# the native code (39 in round 5); the ones that select a route are all here, the rest is tuning / diagnostics
# (MTM_CLASS_LANES, MTM_UPLOAD_BANDS, MTM_BAND_MIN_FILL, MTM_GROUP_SPIN_US, MTM_COMM_TIMEOUT_S, MTM_HOST_TRACE).
DEFAULT_MODES="X=0 MTM_FUSE_LAYOUT=0 MTM_CAND_PINNED=0 MTM_SEG_SKIP=0 MTM_ROW_MUX=0 MTM_HITS_ONLY=0 MTM_EXACT_DIV=0 MTM_EXACT_DIV=2 MTM_FUSE_STATS=0 MTM_KERNEL=dot4 MTM_MFMA_R2=0 MTM_SCREEN_L1=0 MTM_F32_MFMA=2 MTM_F32_MFMA=3 MTM_F32_MFMA=0 MTM_TEMPL_ON_DEVICE=0 MTM_UPLOAD_BANDS=1 MTM_CLASS_LANES=1 MTM_CLASS_LANES=4 MTM_BAND_MIN_FILL=0 MTM_MASKSQ_FUSED=0 MTM_SPARSE_MAPS=0 MTM_NMS_DEVICE_MIN=-1"
# ALT_MODES: a subset of the switches (space separated) instead of all of them
for e in ${ALT_MODES:-$DEFAULT_MODES}; do
  # ALT_K: optional pytest -k expression for a quick pass (e.g. ALT_K="not cfg" tools/alt_modes.sh)
  echo "== $e"; env $e timeout 900 python -m pytest tests -m gpu -q ${ALT_K:+-k "$ALT_K"} 2>&1 | grep -E "passed|failed|error" | tail -2
done
