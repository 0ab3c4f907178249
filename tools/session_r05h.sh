set -u
O=gpurun_out/r05h; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "mask or cfg5" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3; grep -E "^E  " $O/pytest.log | head -10
for i in 1 2; do for e in 1 0; do
  MTM_MASKSQ_RUNS=$e python bench.py --config cfg5 --no-cpu-baseline --skip-extras --steps 10 --warmup 3 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('masksq_runs=$e', d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], d.get('gpu_ms'), r.get('masked_stat'))" | tee -a $O/cfg5_ab.txt
done; done
