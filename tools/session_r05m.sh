set -u
O=gpurun_out/r05m; mkdir -p $O
for e in "X=0" "MTM_CAND_PINNED=0" "MTM_ZERO_IN_STATS=0" "MTM_FUSE_LAYOUT=0" "MTM_BAND_ALIGN=0" "MTM_EAGER_COPY_STREAM=0" "MTM_EXACT_DIV=0" "MTM_NMS_DEVICE=0"; do
  env $e python bench.py --config cfg3 --steps 20 --warmup 3 --no-cpu-baseline --skip-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 $e', d['ms_per_step'], d.get('median_ms_per_call'), d['gpu_ms'])" | tee -a $O/cfg3.txt
done
MTM_HOST_TRACE=1 python bench.py --config cfg3 --steps 20 --warmup 3 --no-cpu-baseline --skip-extras 2>&1 | grep "host trace" | tee $O/cfg3_trace.txt
