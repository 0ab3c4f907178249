#!/bin/bash
mkdir -p gpurun_out/r05x2
python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep "::" | head -101 > /tmp/prefix.txt
timeout 110 python -m pytest -x -q -s -p no:cacheprovider $(cat /tmp/prefix.txt) tools/probes/diag_u8_test.py 2>&1 | grep -E "DIAG8|passed|failed|Error" | cut -c1-300 > gpurun_out/r05x2/diag_u8.txt
cat gpurun_out/r05x2/diag_u8.txt
