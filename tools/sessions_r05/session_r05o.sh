#!/bin/bash
mkdir -p gpurun_out/r05o
{
python tools/probes/u16_many_probe.py
for e in MTM_HITS_ONLY=0 MTM_KPACK=0 MTM_SCREEN_L1=0 MTM_EXACT_DIV=0 MTM_CAND_PINNED=0 MTM_CLASS_LANES=1 MTM_FUSE_PEAKS=0 MTM_UPLOAD_BANDS=1 MTM_CAND_STAGE=0; do
  env $e U16_TAG=$e U16_REPS=2 python tools/probes/u16_many_probe.py 2>&1 | grep -v "^   \|^ \[\|^  \[" 
done
} > gpurun_out/r05o/u16_many.txt 2>&1
cat gpurun_out/r05o/u16_many.txt | cut -c1-400
