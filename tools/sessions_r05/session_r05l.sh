#!/bin/bash
mkdir -p gpurun_out/r05l
for k in 1 2 3; do
  FUZZ_DTYPE=float32 FUZZ_VS_EXACT=1 timeout 300 python tools/fuzz_parity.py 3000 250 > gpurun_out/r05l/f32_$k.txt 2>&1; tail -2 gpurun_out/r05l/f32_$k.txt
  timeout 300 python tools/fuzz_parity.py 7000 250 > gpurun_out/r05l/all_$k.txt 2>&1; grep -E "REAL|fuzz:" gpurun_out/r05l/all_$k.txt
done
