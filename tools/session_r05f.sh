set -u
O=gpurun_out/r05f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
for i in 1 2 3; do for e in 1 0; do
  MTM_SINGLE_BAND=$e python bench.py --config cfg2 --steps 300 --warmup 5 --no-cpu-baseline --skip-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 single_band=$e', d['ms_per_step'], d.get('median_ms_per_call'), d['roofline']['kernel_ms_per_step'])" | tee -a $O/cfg2.txt
done; done
python bench.py 2>/dev/null | tail -1 > $O/bench.json
python -c "
import json
d=json.load(open('$O/bench.json'))
print('bench', d['ms_per_step'], d['median_ms_per_call'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])
for k in ('fresh_templates','pinned_image','reciprocal_normalisation','resident_inputs','image_stream','photograph_like_image','large_template_414x400_over_2048x2048'):
    print(k, json.dumps(d.get(k))[:260])
" | tee $O/bench.txt
