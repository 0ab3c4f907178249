set -u
export ENV_CASES="X=0;MTM_BAND_ALIGN=0;MTM_NCC_EVENTS=0;MTM_UPLOAD_BANDS=0.28,0.6,1;MTM_UPLOAD_BANDS=0.28,0.6,1 MTM_BAND_STREAMS=2;MTM_BAND_INLINE=1;MTM_UPLOAD_BANDS=0.33,1" ENV_REPS=4 LIB_STEPS=200
bash tools/gpu_session.sh r05c tests env_cases
O=gpurun_out/r05c
python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "error_bound_holds" 2>&1 | grep -E "float32 bound|passed|failed" > $O/bound.txt
for i in 1 2 3; do
  python tools/probes/group_calls.py 0 200 2>/dev/null | tail -1 >> $O/group.txt
  python tools/probes/group_calls.py 1 200 2>/dev/null | tail -1 >> $O/group.txt
  MTM_GROUP_SPIN_US=0 python tools/probes/group_calls.py 1 200 2>/dev/null | tail -1 | sed 's/^/spin0 /' >> $O/group.txt
  python tools/probes/loop_calls.py 0 200 2>/dev/null | tail -1 >> $O/group.txt
done
BENCH_GROUP_SINGLE=1 python bench.py --gpus 1 --steps 200 --warmup 3 --no-cpu-baseline --skip-extras 2>/dev/null | tail -1 > $O/group1.json
python bench.py --config cfg2 --steps 200 --warmup 5 --no-cpu-baseline --skip-extras 2>/dev/null | tail -1 > $O/cfg2.json
python -c "
import json
for f in ('cfg2','group1'):
    d=json.load(open('$O/%s.json'%f)); print(f, d['ms_per_step'], d.get('median_ms_per_call'), d['roofline']['kernel_ms_per_step'])
" | tee $O/extra.txt
export TL_CASES="group X=0" TL_TAIL=30 TL_CMD="tools/probes/group_calls.py 1"
bash tools/probes/timeline.sh r05c
cat $O/bound.txt $O/group.txt
