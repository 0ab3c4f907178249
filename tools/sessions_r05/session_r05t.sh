#!/bin/bash
mkdir -p gpurun_out/r05t2
python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep "::" | head -101 > /tmp/prefix.txt
T=tests/test_gpu_parity.py::test_uint16_many_templates
{
for e in MTM_FUSE_STATS=0 MTM_KPACK=0 MTM_KERNEL=naive; do
  echo "== prefix 101 + target, $e"; env $e python -m pytest -x -q -p no:cacheprovider $(cat /tmp/prefix.txt) $T 2>&1 | grep -E "passed|failed|^FAILED|AssertionError: \(" | head -4
done
} > gpurun_out/r05t2/which.txt 2>&1
cat gpurun_out/r05t2/which.txt
