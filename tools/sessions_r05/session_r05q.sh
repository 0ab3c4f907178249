#!/bin/bash
mkdir -p gpurun_out/r05q2
{
for e in X=0 ALT_GEOM=1 MTM_CLASS_LANES=1 MTM_CAND_PINNED=0 MTM_HITS_ONLY=0 MTM_SCREEN_L1=0 MTM_CAND_STAGE=0 "ALT_DTYPE=uint8" "ALT_DTYPE=uint8 ALT_GEOM=1"; do
  env $e ALT_TAG="$e" python tools/probes/u16_alternate.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"
done
} > gpurun_out/r05q2/u16_alternate.txt 2>&1
cat gpurun_out/r05q2/u16_alternate.txt | cut -c1-330
