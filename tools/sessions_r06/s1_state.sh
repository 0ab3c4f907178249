# Round 6, second session, call 1: where the tree stands - suite, the three route switches that failed in the first alt-mode
# run (with their failure text), masked float32 probe (lists + N_object == 1), default bench line.
set -u
O=gpurun_out/r06s1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; tail -3 $O/pytest_all.log
for e in MTM_HITS_ONLY=0 MTM_KERNEL=dot4 MTM_F32_MFMA=2; do
  echo "== $e"; env $e timeout 900 python -m pytest tests -m gpu -x -q > $O/alt_$e.log 2>&1; tail -40 $O/alt_$e.log | grep -E "^E|^FAILED|^ERROR|passed|failed" | head -30
done
timeout 600 python tools/probes/maskf32_probe.py > $O/maskf32_probe.txt 2>&1; tail -8 $O/maskf32_probe.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['kernel_ms_per_step'], d['fresh_templates']['median_ms_per_call'], d['score_maps_materialised']['ncc_kernel_ms'], d['photograph_like_image']['median_ms_per_call'])"
