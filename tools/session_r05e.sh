set -u
O=gpurun_out/r05e; mkdir -p $O
for i in 1 2; do
  for e in 1 0; do
    MTM_EAGER_COPY_STREAM=$e python tools/probes/rccl_first_probe.py 200 2>/dev/null | grep "communicator first" | tee -a $O/rccl_first.txt
    MTM_EAGER_COPY_STREAM=$e python tools/probes/group_calls.py 1 200 2>/dev/null | grep "group of" | sed "s/^/eager=$e /" | tee -a $O/group.txt
  done
  python tools/probes/group_calls.py 0 200 2>/dev/null | grep "group of" | tee -a $O/group.txt
  python tools/probes/group_calls.py 2 200 2>/dev/null | grep "group of" | sed "s/^/comm alive, host merge: /" | tee -a $O/group.txt
  python tools/probes/loop_calls.py 0 200 2>/dev/null | tail -1 | tee -a $O/group.txt
done
BENCH_GROUP_SINGLE=1 python bench.py --gpus 1 --steps 200 --warmup 3 --no-cpu-baseline --skip-extras 2>/dev/null | tail -1 > $O/group1.json
python -c "
import json
d=json.load(open('$O/group1.json')); print('group1', d['ms_per_step'], d.get('median_ms_per_call'), d['roofline']['kernel_ms_per_step'], d.get('multi_gpu'))
" | tee $O/extra.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "group or nms or dense or sparse or fused" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
