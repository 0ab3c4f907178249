#!/bin/bash
mkdir -p gpurun_out/r05n
python -m pytest tests -m gpu -x -q > gpurun_out/r05n/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r05n/pytest.log
python bench.py > gpurun_out/r05n/bench.json 2>gpurun_out/r05n/bench.err
MTM_SEG_SKIP=0 python bench.py > gpurun_out/r05n/bench_noskip.json 2>>gpurun_out/r05n/bench.err
python - <<'PY'
import json
for f in ("bench", "bench_noskip"):
    d=json.loads(open("gpurun_out/r05n/%s.json"%f).read().strip().splitlines()[-1])
    e=d["extras"] if "extras" in d else d
    print(f, d["ms_per_step"], d["roofline"]["frac"], json.dumps(e.get("photograph_like_image"))[:260])
PY
python tools/probes/dense_probe.py 0.5 0.7 > gpurun_out/r05n/dense.txt 2>&1; grep -E "call median|dense call" gpurun_out/r05n/dense.txt
MTM_SEG_SKIP=0 python tools/probes/dense_probe.py 0.5 0.7 > gpurun_out/r05n/dense_noskip.txt 2>&1; grep -E "call median|dense call" gpurun_out/r05n/dense_noskip.txt
