#!/bin/bash
mkdir -p gpurun_out/r05r2
{
echo "== alone"; python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "uint16_many" 2>&1 | grep -E "passed|failed"
for e in X=0 MTM_CAND_PINNED=0 MTM_CLASS_LANES=1 MTM_SCREEN_L1=0 MTM_ZERO_IN_STATS=0; do
  echo "== whole suite, $e"; env $e python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|AssertionError: \(" | head -5
done
} > gpurun_out/r05r2/bisect.txt 2>&1
cat gpurun_out/r05r2/bisect.txt
