#!/bin/bash
mkdir -p gpurun_out/r05p2
{
for e in X=0 MTM_CAND_PINNED=0 MTM_CLASS_LANES=1 MTM_SCREEN_L1=0 MTM_CAND_STAGE=0 MTM_KPACK=0 MTM_HITS_ONLY=0 MTM_FUSE_PEAKS=0 X=1; do
  env $e U16_TAG=$e python tools/probes/u16_stress.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"
done
} > gpurun_out/r05p2/u16_stress.txt 2>&1
cat gpurun_out/r05p2/u16_stress.txt | cut -c1-300
