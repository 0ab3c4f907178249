#!/bin/bash
mkdir -p gpurun_out/r05v2
python -m pytest tests -m gpu -x -q > gpurun_out/r05v2/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r05v2/pytest.log
python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep "::" | head -101 > /tmp/prefix.txt
python -m pytest -x -q -s -p no:cacheprovider $(cat /tmp/prefix.txt) tools/probes/diag_u16_test.py 2>&1 | grep -E "DIAG|passed|failed" | cut -c1-400 > gpurun_out/r05v2/diag.txt
cat gpurun_out/r05v2/diag.txt
python tools/probes/f32_probe.py 2>/dev/null | grep -E "uint16|uint8" > gpurun_out/r05v2/u16_perf.txt; cat gpurun_out/r05v2/u16_perf.txt
timeout 200 python tools/fuzz_parity.py 7000 250 2>&1 | tail -2 > gpurun_out/r05v2/fuzz.txt; cat gpurun_out/r05v2/fuzz.txt
