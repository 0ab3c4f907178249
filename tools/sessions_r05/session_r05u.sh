#!/bin/bash
mkdir -p gpurun_out/r05u2
python -m pytest tests -m gpu -x -q > gpurun_out/r05u2/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r05u2/pytest.log
python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep "::" | head -101 > /tmp/prefix.txt
MTM_LIB_PATH=$PWD/multitemplatematching-python_amd/MTM/libmtm_hip_detector.so python -m pytest -x -q -s -p no:cacheprovider $(cat /tmp/prefix.txt) tools/probes/diag_u16_test.py 2>&1 | grep -E "DIAG|passed|failed" | cut -c1-900 > gpurun_out/r05u2/diag.txt
cat gpurun_out/r05u2/diag.txt
