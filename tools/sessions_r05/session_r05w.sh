#!/bin/bash
mkdir -p gpurun_out/r05w2
python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep "::" | head -101 > /tmp/prefix.txt
for tag in detfix base; do
  echo "== $tag" >> gpurun_out/r05w2/diag.txt
  MTM_LIB_PATH=$PWD/multitemplatematching-python_amd/MTM/libmtm_hip_$tag.so python -m pytest -x -q -s -p no:cacheprovider $(cat /tmp/prefix.txt) tools/probes/diag_u16_test.py 2>&1 | grep -E "DIAG|passed|failed" | cut -c1-330 >> gpurun_out/r05w2/diag.txt
done
cat gpurun_out/r05w2/diag.txt
