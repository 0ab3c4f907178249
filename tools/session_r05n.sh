set -u
O=gpurun_out/r05n; mkdir -p $O
for rep in 1 2 3; do for cfg in cfg2 cfg3 cfg4 cfg5; do
  python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench_configs_rep$rep.jsonl
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05n/bench_configs_rep*.jsonl')):
    print(f[-11:], [ (json.loads(l)['ms_per_step'], json.loads(l)['roofline']['kernel_ms_per_step']) for l in open(f)])
PY
