#!/bin/bash
# One gpurun call = one GPU box for a few minutes: run a list of stages back to back and leave everything under
# gpurun_out/<tag>/ (merged back into the repo by gpurun).  Usage: tools/gpu_session.sh <tag> stage [stage ...]
# Stages: tests | bench | screen_ab | f32 | fuzz_f32 | profile | alt_quick | group | probes
set -u
TAG=${1:-s}; shift
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
clean() { grep -vE "^RCCL|^HIP|^ROCm|^Host|^Librccl" ; }
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $OUT/session.log; }
for st in "$@"; do
  stamp "stage $st"
  case $st in
    tests)
      timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; stamp "pytest rc=$? $(tail -1 $OUT/pytest.log)" ;;
    tests_all)      # no -x: every failure listed
      timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_all.log 2>&1; stamp "pytest rc=$? $(tail -1 $OUT/pytest_all.log)" ;;
    bench)
      python bench.py 2>$OUT/bench.err | clean | tail -1 > $OUT/bench.json; stamp "bench $(cut -c1-160 $OUT/bench.json)" ;;
    bench_driver)   # the driver's command line
      python bench.py --gpus 1 --steps 20 --warmup 5 2>>$OUT/bench.err | clean | tail -1 > $OUT/bench_driver.json; stamp "bench_driver $(cut -c1-160 $OUT/bench_driver.json)" ;;
    screen_ab)      # the per-lane bound of the hits-only screen on / off, same box, alternating
      for rep in 1 2; do for v in 1 0; do
        MTM_SCREEN_L1=$v python bench.py --no-cpu-baseline --skip-extras --steps 200 2>>$OUT/bench.err | clean | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('L1=$v', d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], r.get('sclk_mhz_in_kernel'))" | tee -a $OUT/screen_ab.txt
      done; done ;;
    f32)
      for v in 1 2 0; do echo "== MTM_F32_MFMA=$v" >> $OUT/f32_probe.txt; MTM_F32_MFMA=$v timeout 300 python tools/probes/f32_probe.py >> $OUT/f32_probe.txt 2>&1; done
      stamp "$(grep float32 $OUT/f32_probe.txt | tr '\n' '|')" ;;
    fuzz_f32)
      FUZZ_DTYPE=float32 FUZZ_VS_EXACT=1 timeout 600 python tools/fuzz_parity.py 0 ${FUZZ_N:-500} > $OUT/fuzz_f32.txt 2>&1; stamp "fuzz_f32: $(tail -2 $OUT/fuzz_f32.txt | tr '\n' ' ')" ;;
    fuzz)
      timeout 420 python tools/fuzz_parity.py 1000 ${FUZZ_N:-150} > $OUT/fuzz.txt 2>&1; stamp "fuzz: $(tail -2 $OUT/fuzz.txt | tr '\n' ' ')" ;;
    profile)
      bash tools/profile_round.sh ${PROFILE_TAG:-$TAG/profile} > $OUT/profile.log 2>&1; stamp "profile done" ;;
    alt_quick)
      ALT_K="${ALT_K:-not cfg and not full_size}" bash tools/alt_modes.sh > $OUT/alt_modes.txt 2>&1; stamp "alt: $(grep -c passed $OUT/alt_modes.txt) modes, $(grep -c failed $OUT/alt_modes.txt) with failures" ;;
    configs)        # the BASELINE configs on one GPU (roofline.traffic from the committed PMC table)
      for cfg in cfg2 cfg3 cfg4 cfg5; do python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline 2>>$OUT/bench.err | clean | tail -1 >> $OUT/bench_configs.jsonl; done
      stamp "configs: $(wc -l < $OUT/bench_configs.jsonl) lines" ;;
    group)
      BENCH_GROUP_ALIAS=1 python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline 2>>$OUT/bench.err | clean | tail -1 > $OUT/bench_group2.json
      BENCH_GROUP_ALIAS=1 python bench.py --gpus 8 --steps 10 --warmup 2 --no-cpu-baseline 2>>$OUT/bench.err | clean | tail -1 > $OUT/bench_group8.json
      BENCH_GROUP_SINGLE=1 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --skip-extras 2>>$OUT/bench.err | clean | tail -1 > $OUT/bench_group1_rccl.json
      stamp "group: $(cut -c1-120 $OUT/bench_group2.json) | $(cut -c1-120 $OUT/bench_group8.json) | $(python -c "import json;print(json.load(open('$OUT/bench_group1_rccl.json'))['multi_gpu'])" 2>&1 | cut -c1-200)" ;;
    probes)
      for pr in ${PROBES:-call_breakdown dense_probe coins_probe}; do timeout 300 python tools/probes/$pr.py > $OUT/$pr.txt 2>&1; stamp "$pr: $(tail -3 $OUT/$pr.txt | tr '\n' '|' | cut -c1-300)"; done ;;
    bands_ab)       # upload-band layouts x one / two score streams, per-call metric
      for d in 0 1; do for b in "0.25,1" "0.1,0.3,0.6,1" "0.08,0.22,0.46,0.9,1" "0.12,0.34,0.56,0.78,1"; do
        MTM_DUAL_STREAM=$d MTM_UPLOAD_BANDS="$b" python bench.py --steps 150 --warmup 5 --skip-extras --no-cpu-baseline 2>>$OUT/bench.err | clean | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('dual=$d bands=$b', d['ms_per_step'], d['median_ms_per_call'], r['kernel_ms_per_step'], r['launches_per_step'], d['gpu_ms'])" | tee -a $OUT/bands_ab.txt
      done; done ;;
    rows_ab)        # three-row (default) against two-row MFMA variant, banded and single-launch, alternating
      for rep in 1 2; do for v in 1 3; do for b in "0.25,1" "1"; do
        MTM_MFMA_R2=$v MTM_UPLOAD_BANDS="$b" python bench.py --no-cpu-baseline --skip-extras --steps 200 2>>$OUT/bench.err | clean | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('R2=$v bands=$b', d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], r.get('sclk_mhz_in_kernel'))" | tee -a $OUT/rows_ab.txt
      done; done; done ;;
    lib_ab)         # experiment builds (MTM_BUILD_TAG=<tag> -> libmtm_hip_<tag>.so) against the product library, same box, alternating
      L=multitemplatematching-python_amd/MTM
      for rep in 1 2 3; do for v in ${LIB_TAGS:-v0 v2} ""; do
        lib=$R/$L/libmtm_hip${v:+_$v}.so; [ -f $lib ] || continue
        for b in ${LIB_BANDS:-"0.25,1" "1"}; do
        MTM_LIB_PATH=$lib MTM_UPLOAD_BANDS="$b" python bench.py --no-cpu-baseline --skip-extras --steps ${LIB_STEPS:-200} 2>>$OUT/bench.err | clean | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('lib=${v:-product} bands=$b', d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], r.get('sclk_mhz_in_kernel'))" | tee -a $OUT/lib_ab.txt
        done
      done; done ;;
    persist_ab)     # item scheduling of the score kernel: launch per item / persistent (global or per-XCD draw), with and without the start stagger
      for rep in 1 2; do for v in "0 -1 0" "2 -1 0" "2 0 0" "1 -1 0" "0 -1 3" "2 2 0" "2 6 0"; do set -- $v
        for b in ${LIB_BANDS:-"0.25,1" "1"}; do
        MTM_MFMA_PERSISTENT=$1 MTM_MFMA_STAGGER=$2 MTM_MFMA_STAGGER_NP=$3 MTM_UPLOAD_BANDS="$b" python bench.py --no-cpu-baseline --skip-extras --steps ${LIB_STEPS:-200} 2>>$OUT/bench.err | clean | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('persistent=$1 stagger=$2 stagger_np=$3 bands=$b', d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], r.get('sclk_mhz_in_kernel'))" | tee -a $OUT/persist_ab.txt
        done
      done; done ;;
    cfg5_ab)        # sum I^2 M of the masked classes: one fused launch per class against round 3's two raw launches + combine
      for rep in 1 2; do for v in 1 0; do
        MTM_MASKSQ_FUSED=$v python bench.py --config cfg5 --no-cpu-baseline --skip-extras --steps 10 --warmup 3 2>>$OUT/bench.err | clean | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('masksq_fused=$v', d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], d.get('gpu_ms'), r.get('masked_stat'))" | tee -a $OUT/cfg5_ab.txt
      done; done ;;
    rm_ab)          # row-multiplexed tilings: one-group edge steps on / off (few-template calls, slabs)
      for v in 1 0 1 0; do MTM_RM_EDGES=$v timeout 300 python tools/probes/rm_probe.py 2>&1 | grep -E "x  ?[1248] templates" | sed "s/^/RM_EDGES=$v /" >> $OUT/rm_ab.txt; done
      for v in 1 0; do MTM_RM_EDGES=$v timeout 120 python tools/probes/workload.py slab_414 30 2>&1 | tail -2 | sed "s/^/RM_EDGES=$v /" >> $OUT/rm_ab.txt; done
      stamp "rm_ab: $(grep -c templates $OUT/rm_ab.txt) lines" ;;
    trace)          # host time stamps of the phases of a fused call (stderr of the context at exit) + per-call breakdown
      MTM_HOST_TRACE=1 timeout 120 python tools/probes/loop_calls.py 0 200 > $OUT/host_trace.txt 2>&1
      timeout 120 python tools/probes/call_breakdown.py >> $OUT/host_trace.txt 2>&1
      stamp "trace: $(grep -c 'host trace' $OUT/host_trace.txt) phases" ;;
    timeline)       # rocprofv3 kernel + copy timelines of the fused call (TL_CASES, tools/probes/timeline.sh)
      bash tools/probes/timeline.sh $TAG > $OUT/timeline.log 2>&1; stamp "timeline: $(ls $OUT/timeline_*.csv 2>/dev/null | wc -l) cases" ;;
    wl_ab)          # a named workload (tools/probes/workload.py) under environment switches: WL_NAME=slab_414 WL_ENVS="X=0;MTM_SLAB_CW=128;MTM_SLAB_CW=64 MTM_SLAB_STREAMS=8"
      IFS=';' read -ra WLE <<< "${WL_ENVS:-X=0}"
      for rep in 1 2; do for e in "${WLE[@]}"; do
        env $e timeout 200 python tools/probes/workload.py ${WL_NAME:-slab_414} 30 2>>$OUT/bench.err | clean | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${WL_NAME:-slab_414}', '$e', 'call', d['median_ms_per_call'], 'gpu', d['gpu_ms'], 'ncc', d['ncc_kernel_ms'], 'hits', d['hits'])" | tee -a $OUT/wl_ab.txt
      done; done ;;
    env_cases)      # the bench workload per call under combinations of switches: ENV_CASES="X=0;MTM_A=1 MTM_B=2;..." (ENV_REPS rounds, alternating)
      IFS=';' read -ra ECS <<< "${ENV_CASES:-X=0}"
      for rep in $(seq 1 ${ENV_REPS:-2}); do for e in "${ECS[@]}"; do
        env $e python bench.py --no-cpu-baseline --skip-extras --steps ${LIB_STEPS:-200} 2>>$OUT/bench.err | clean | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$e |', d['ms_per_step'], d['median_ms_per_call'], r['kernel_ms_per_step'], r['frac'])" | tee -a $OUT/env_cases.txt
      done; done ;;
    env_ab)         # any environment switch against the default, per-call metric: ENV_AB="MTM_BAND_STREAMS=1 MTM_CAND_STAGE=0"
      for rep in 1 2 3; do for e in "X=0" ${ENV_AB:-MTM_BAND_STREAMS=1}; do
        env $e python bench.py --no-cpu-baseline --skip-extras --steps ${LIB_STEPS:-200} 2>>$OUT/bench.err | clean | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$e', d['ms_per_step'], d['median_ms_per_call'], r['kernel_ms_per_step'], r['frac'], r.get('sclk_mhz_in_kernel'))" | tee -a $OUT/env_ab.txt
      done; done ;;
    dense_ab)       # the dense regime (photograph-like 4K image x 32 templates, threshold 0.5) under environment switches
      for e in "X=0" ${ENV_AB:-MTM_DENSE_ROWMAX=0 MTM_CAND_STAGE=0}; do
        env $e timeout 200 python tools/probes/dense_probe.py 0.5 2>&1 | grep -A1 "HITS_ONLY=1\|dense call" | sed "s/^/$e /" | cut -c1-360 >> $OUT/dense_ab.txt
      done; stamp "dense_ab: $(grep -c 'call median' $OUT/dense_ab.txt) runs" ;;
    emit_probe)     # where the time of listing ~1e6 candidates goes: hits-only mode with a list that holds them, emission stages switched off one at a time (results invalid, timing only)
      for e in "MTM_MFMA_DBG=0" "MTM_MFMA_DBG=4" "MTM_MFMA_DBG=8" "MTM_MFMA_DBG=16" "MTM_MFMA_DBG=24" "MTM_CAND_STAGE=0"; do
        env DENSE_HIT_CAP=1048576 $e timeout 200 python tools/probes/dense_probe.py 0.5 2>&1 | grep "HITS_ONLY=1" | sed "s/^/$e /" | cut -c1-200 >> $OUT/emit_probe.txt
      done; stamp "emit_probe: $(grep -c 'call median' $OUT/emit_probe.txt) runs" ;;
    bands_fine)     # the first band's share of the rows, finely (per-call metric; alternating)
      for rep in 1 2; do for b in ${BANDS_FINE:-"0.25,1" "0.235,1" "0.265,1" "0.28,1"}; do
        MTM_UPLOAD_BANDS="$b" python bench.py --no-cpu-baseline --skip-extras --steps 200 2>>$OUT/bench.err | clean | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bands=$b', d['ms_per_step'], d['median_ms_per_call'], r['kernel_ms_per_step'], r['frac'])" | tee -a $OUT/bands_fine.txt
      done; done ;;
    lone_wave)      # one work-group per CU (persistent launch): what a wave achieves without a partner on its SIMD
      for e in "MTM_MFMA_PER_CU=2" "MTM_MFMA_PER_CU=1" "MTM_MFMA_PER_CU=1 MTM_MFMA_DBG=2" "MTM_MFMA_PER_CU=2 MTM_MFMA_DBG=2"; do
        env MTM_MFMA_PERSISTENT=1 MTM_MFMA_STAGGER=0 MTM_UPLOAD_BANDS=1 $e python bench.py --no-cpu-baseline --skip-extras --steps 100 2>>$OUT/bench.err | clean | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$e', d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], r.get('sclk_mhz_in_kernel'))" | tee -a $OUT/lone_wave.txt
      done ;;
    ubench)         # prebuilt micro-benchmarks (tools/ubench/<name>/ub)
      for u in ${UBENCH:-step}; do echo "== $u" >> $OUT/ubench.txt; timeout 120 tools/ubench/$u/ub >> $OUT/ubench.txt 2>&1; done
      stamp "ubench: $(grep -c cycles $OUT/ubench.txt) lines" ;;
    workloads)      # rocprofv3 summaries + PMC for the secondary workloads
      bash tools/profile_workloads.sh ${PROFILE_TAG:-$TAG} ${WORKLOADS:-} > $OUT/workloads.log 2>&1; stamp "workloads done: $(grep -c '^==' $OUT/workloads.log)" ;;
    *) stamp "unknown stage $st" ;;
  esac
done
stamp "done"
cat $OUT/session.log
