set -u
O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^FAILED" $O/pytest.log | head
python bench.py --no-cpu-baseline --skip-extras --steps 200 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['median_ms_per_call'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" | tee $O/bench.txt
