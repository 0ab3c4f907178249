#!/bin/bash
# One GPU-box pass that produces everything profiles/ holds for a round: the default bench line, the
# other BASELINE configs, the rocprofv3 kernel-trace summary of the default bench command and the PMC
# passes (each in its own rocprofv3 run, --kernel-trace only).  Usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-rXX}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
clean() { grep -vE "^RCCL|^HIP|^ROCm|^Host|^Librccl" ; }
python bench.py 2>$OUT/bench.err | clean | tail -1 > $OUT/bench.json
for cfg in cfg2 cfg3 cfg5; do
  python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline 2>>$OUT/bench.err | clean | tail -1 >> $OUT/bench_configs.jsonl
done
MTM_HITS_ONLY=0 python bench.py --no-cpu-baseline 2>>$OUT/bench.err | clean | tail -1 > $OUT/bench_maps_materialised.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python $R/bench.py --no-cpu-baseline --skip-extras > $OUT/prof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB > $OUT/kernel_stats.csv
export BENCH_PREWARM=4
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extras"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o $name -- $CMD > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed"
        db=$(find $OUT/pmc_$name -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db > $OUT/pmc_$name.csv; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU
run p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run p3 FETCH_SIZE GRBM_GUI_ACTIVE
run p4 WRITE_SIZE GRBM_GUI_ACTIVE
HO="env MTM_HITS_ONLY=0"; CMD="$HO python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extras"
run p3_maps FETCH_SIZE GRBM_GUI_ACTIVE
run p4_maps WRITE_SIZE GRBM_GUI_ACTIVE
unset BENCH_PREWARM
cd $R
rm -rf $OUT/prof $OUT/pmc_p1 $OUT/pmc_p2 $OUT/pmc_p3 $OUT/pmc_p4 $OUT/pmc_p3_maps $OUT/pmc_p4_maps
ls -la $OUT; cat $OUT/bench.json | cut -c1-1500; echo; head -8 $OUT/kernel_stats.csv
