#!/bin/bash
mkdir -p gpurun_out/r05h
P="python tools/probes/fuzz_case.py 7169"
{
FUZZ_TAG=default $P
FUZZ_TAG=only_t27 FUZZ_ONLY=t27 $P
FUZZ_TAG=only_t26_t27 FUZZ_ONLY=t26,t27 $P
for e in MTM_KERNEL=naive MTM_KERNEL=dot4 MTM_MFMA_R2=0 MTM_HITS_ONLY=0 MTM_FUSE_PEAKS=0 MTM_ROW_MUX=0 MTM_KPACK=0 MTM_FUSE_STATS=0 MTM_SINGLE_BAND=0 MTM_FUSE_LAYOUT=0 MTM_EXACT_DIV=0 MTM_SCREEN_L1=0 MTM_CAND_PINNED=0 MTM_TEMPL_ON_DEVICE=0; do
  env $e FUZZ_TAG=$e $P
done
} > gpurun_out/r05h/case7169.txt 2>&1
cat gpurun_out/r05h/case7169.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r05h/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r05h/pytest.log
python tools/probes/f32_raw_probe.py > gpurun_out/r05h/f32_raw.txt 2>&1; grep cfg gpurun_out/r05h/f32_raw.txt
