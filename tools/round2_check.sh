#!/bin/bash
# GPU box: parity suite + bench lines (default and upload-band variants).  Logs under gpurun_out/r02a/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r02a
mkdir -p $out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout 600 python bench.py --steps 40 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
echo "bench rc=$?"; tail -c 600 $out/bench_default.json
for b in "1" "0.25,1" "0.12,0.4,0.7,1" "0.08,0.3,0.53,0.76,1" "0.1,0.25,0.4,0.55,0.7,0.85,1"; do
  MTM_UPLOAD_BANDS="$b" timeout 300 python bench.py --steps 60 --warmup 5 --skip-extras --no-cpu-baseline 2>> $out/bands.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bands=$b', d['value'], d['ms_per_step'], d['median_ms_per_call'], d['roofline']['kernel_ms_per_step'], d['roofline']['launches_per_step'], d['clock'])" | tee -a $out/bands.log
done
