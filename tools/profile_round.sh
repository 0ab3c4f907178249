#!/bin/bash
# One GPU-box pass that produces everything profiles/ holds for a round: the default bench line, the other
# BASELINE configs, rocprofv3 kernel-trace summaries of the bench command (as shipped: the image arrives in row
# bands, one score launch per band; and with MTM_UPLOAD_BANDS=1: one full-image launch per step) and the PMC
# passes (each in its own rocprofv3 run, --kernel-trace only, full-image launches so that "per launch" means
# the whole 4K x 32 workload).  Usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-rXX}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
clean() { grep -vE "^RCCL|^HIP|^ROCm|^Host|^Librccl" ; }
python bench.py 2>$OUT/bench.err | clean | tail -1 > $OUT/bench.json
for cfg in cfg2 cfg3 cfg4 cfg5; do
  python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline 2>>$OUT/bench.err | clean | tail -1 >> $OUT/bench_configs.jsonl
done
MTM_HITS_ONLY=0 python bench.py --no-cpu-baseline --skip-extras 2>>$OUT/bench.err | clean | tail -1 > $OUT/bench_maps_materialised.json
cd /tmp
prof() { name=$1; shift
  timeout 600 env "$@" rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o prof -- python $R/bench.py --no-cpu-baseline --skip-extras > $OUT/prof_$name.log 2>&1
  DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB > $OUT/kernel_stats_$name.csv
  grep -h '"metric"' $OUT/prof_$name.log | tail -1 > $OUT/bench_under_rocprof_$name.json; }
prof banded MTM_X=0
prof single MTM_UPLOAD_BANDS=1
export BENCH_PREWARM_S=0.05 MTM_UPLOAD_BANDS=1
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extras"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o $name -- $CMD > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed"
        db=$(find $OUT/pmc_$name -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db > $OUT/pmc_$name.csv; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU
run p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run p3 FETCH_SIZE GRBM_GUI_ACTIVE
run p4 WRITE_SIZE GRBM_GUI_ACTIVE
run p5 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE
export MTM_HITS_ONLY=0
run p3_maps FETCH_SIZE GRBM_GUI_ACTIVE
run p4_maps WRITE_SIZE GRBM_GUI_ACTIVE
unset BENCH_PREWARM_S MTM_UPLOAD_BANDS MTM_HITS_ONLY
cd $R
rm -rf $OUT/prof_banded $OUT/prof_single $OUT/pmc_p1 $OUT/pmc_p2 $OUT/pmc_p3 $OUT/pmc_p4 $OUT/pmc_p5 $OUT/pmc_p3_maps $OUT/pmc_p4_maps
ls -la $OUT; cut -c1-600 $OUT/bench.json; echo; head -8 $OUT/kernel_stats_banded.csv; head -5 $OUT/kernel_stats_single.csv; grep ncc_mfma $OUT/pmc_p1.csv $OUT/pmc_p3.csv $OUT/pmc_p4.csv | head -20
