#!/bin/bash
# GPU parity suite under every alternative code path of the library (GPU box)
DEFAULT_MODES="X=0 MTM_FUSE_LAYOUT=0 MTM_CAND_PINNED=0 MTM_ZERO_IN_STATS=0 MTM_BAND_ALIGN=0 MTM_SINGLE_BAND=0 MTM_SEG_SKIP=0 MTM_F32_RIG=0 MTM_ROW_MUX=0 MTM_HITS_ONLY=0 MTM_EXACT_DIV=0 MTM_EXACT_DIV=2 MTM_FUSE_PEAKS=0 MTM_FUSE_STATS=0 MTM_KERNEL=dot4 MTM_MFMA_R2=0 MTM_MFMA_R2=3 MTM_SCREEN_L1=0 MTM_F32_MFMA=2 MTM_TEMPL_ON_DEVICE=0 MTM_UPLOAD_BANDS=1 MTM_SLAB_MFMA=0 MTM_F32_MFMA=0 MTM_KPACK=0 MTM_CLASS_LANES=1 MTM_CLASS_LANES=4 MTM_SLAB_STREAMS=1 MTM_BAND_MIN_FILL=0 MTM_MASKSQ_FUSED=0 MTM_MFMA_PERSISTENT=2 MTM_RM_EDGES=0 MTM_CAND_STAGE=0 MTM_SLAB_CW=128 MTM_SLAB_CW=64 MTM_SLAB_MERGE=0 MTM_SPARSE_MAPS=0 MTM_NMS_DEVICE=0"
# ALT_MODES: a subset of the switches (space separated) instead of all of them
for e in ${ALT_MODES:-$DEFAULT_MODES}; do
  # ALT_K: optional pytest -k expression for a quick pass (e.g. ALT_K="not cfg" tools/alt_modes.sh)
  echo "== $e"; env $e timeout 600 python -m pytest tests -m gpu -x -q ${ALT_K:+-k "$ALT_K"} 2>&1 | grep -E "passed|failed|error" | tail -2
done
