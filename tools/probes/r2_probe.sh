#!/bin/bash
# GPU box: parity suite with the two-row MFMA variant on, then kernel / call timing with it on and off.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for r in 1 0 1 0; do
  MTM_MFMA_R2=$r timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; ri=d['resident_inputs']; print('R2=$r value', d['value'], 'median', d['median_ms_per_call'], 'resident kernel', ri['kernel_ms_per_launch'], 'pipelined', ri['pipelined_ms_per_step'], 'clk', ri['sclk_mhz_in_kernel'], 'maps', d['score_maps_materialised']['ncc_kernel_ms'], d['score_maps_materialised']['identical_hits'], ri['identical_hits'])"
done
