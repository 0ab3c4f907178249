set -u
cd $GRAFT_REPO_ROOT 2>/dev/null || true
export ENV_CASES="X=0;MTM_CAND_PINNED=0;MTM_FUSE_LAYOUT=0;MTM_CAND_PINNED=0 MTM_FUSE_LAYOUT=0;MTM_EXACT_DIV=1" ENV_REPS=2 LIB_STEPS=200
export LIB_TAGS=pb8 LIB_BANDS="0.25,1"
export TL_CASES="default X=0;r4 MTM_CAND_PINNED=0 MTM_FUSE_LAYOUT=0" TL_TAIL=24
bash tools/gpu_session.sh r05a tests env_cases lib_ab timeline trace
python bench.py --config cfg2 --steps 200 --warmup 5 --no-cpu-baseline --skip-extras 2>/dev/null | tail -1 > gpurun_out/r05a/cfg2.json
MTM_CAND_PINNED=0 MTM_FUSE_LAYOUT=0 python bench.py --config cfg2 --steps 200 --warmup 5 --no-cpu-baseline --skip-extras 2>/dev/null | tail -1 > gpurun_out/r05a/cfg2_r4.json
BENCH_GROUP_SINGLE=1 python bench.py --gpus 1 --steps 50 --warmup 3 --no-cpu-baseline --skip-extras 2>/dev/null | tail -1 > gpurun_out/r05a/group1.json
python -c "
import json
for f in ('cfg2','cfg2_r4','group1'):
    d=json.load(open('gpurun_out/r05a/%s.json'%f)); print(f, d['ms_per_step'], d.get('median_ms_per_call'), d['roofline']['kernel_ms_per_step'])
"
