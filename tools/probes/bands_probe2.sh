#!/bin/bash
# GPU box: per-call bench line for upload-band layouts x single / dual score streams.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r02c; mkdir -p $out
for d in 0 1; do
for b in "0.25,1" "0.1,0.4,1" "0.2,0.6,1" "0.12,0.34,0.56,0.78,1" "1"; do
  MTM_DUAL_STREAM=$d MTM_UPLOAD_BANDS="$b" timeout 300 python bench.py --steps 80 --warmup 5 --skip-extras --no-cpu-baseline 2>> $out/bands.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('dual=$d bands=$b', d['value'], d['ms_per_step'], d['median_ms_per_call'], r['kernel_ms_per_step'], r['kernel_ms_per_launch'], r['launches_per_step'], d['clock']['sclk_mhz_in_kernel'])" | tee -a $out/bands.log
done
done
