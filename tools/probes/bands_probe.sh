#!/bin/bash
# GPU box: per-call bench line for several upload-band layouts (MTM_UPLOAD_BANDS) + the fused-call parity test.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r02b; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_image or device_group" 2>&1 | tail -3
for b in "1" "0.25,1" "0.16,0.44,0.72,1" "0.12,0.34,0.56,0.78,1" "0.1,0.25,0.4,0.55,0.7,0.85,1" "0.08,0.2,0.32,0.44,0.56,0.68,0.8,0.9,1"; do
  MTM_UPLOAD_BANDS="$b" timeout 300 python bench.py --steps 60 --warmup 5 --skip-extras --no-cpu-baseline 2>> $out/bands.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bands=$b', d['value'], d['ms_per_step'], d['median_ms_per_call'], r['kernel_ms_per_step'], r['kernel_ms_per_launch'], r['launches_per_step'], d['gpu_ms'], d['clock']['sclk_mhz_in_kernel'])" | tee -a $out/bands.log
done
