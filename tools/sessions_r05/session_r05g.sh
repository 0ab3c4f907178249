#!/bin/bash
# round 5, session g: raw-sum float32 thresholds on the bf16 kernel - suite, probe, fuzz
mkdir -p gpurun_out/r05g
python -m pytest tests -m gpu -x -q > gpurun_out/r05g/pytest.log 2>&1; tail -3 gpurun_out/r05g/pytest.log
python tools/probes/f32_raw_probe.py > gpurun_out/r05g/f32_raw.txt 2>&1; cat gpurun_out/r05g/f32_raw.txt
FUZZ_DTYPE=float32 FUZZ_VS_EXACT=1 timeout 500 python tools/fuzz_parity.py 3000 250 > gpurun_out/r05g/fuzz_f32.txt 2>&1; tail -4 gpurun_out/r05g/fuzz_f32.txt
timeout 300 python tools/fuzz_parity.py 7000 250 > gpurun_out/r05g/fuzz_all.txt 2>&1; tail -3 gpurun_out/r05g/fuzz_all.txt
python bench.py > gpurun_out/r05g/bench.json 2>gpurun_out/r05g/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05g/bench.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["value"], d["roofline"]["frac"])
PY
