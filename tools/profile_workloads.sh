#!/bin/bash
# rocprofv3 evidence for everything that is NOT the headline kernel (GPU box): per workload a kernel-trace summary
# (every kernel of the call with its share) and the PMC passes - SQ instruction mix / MFMA busy, FETCH_SIZE, WRITE_SIZE -
# each in a rocprofv3 run of its own (--kernel-trace only next to --pmc).  Output: gpurun_out/<tag>/wl_<name>_*.csv|json;
# tools/collect_profiles.py copies them to profiles/.   Usage: tools/profile_workloads.sh <tag> [name ...]
set -u
TAG=${1:-rXX}; shift
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
NAMES=${*:-cfg2 cfg3 cfg4 cfg5 u16_4k32 f32_4k32 f64_1080p8 slab_414 dense_4k32 dense_4k32_nms}
cd /tmp
for n in $NAMES; do
  python $R/tools/probes/workload.py $n 2>/dev/null | tail -1 > $OUT/wl_${n}.json
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/wlprof_$n -o prof -- python $R/tools/probes/workload.py $n 6 > $OUT/wl_${n}_prof.log 2>&1
  DB=$(find $OUT/wlprof_$n -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py "$DB" > $OUT/wl_${n}_kernel_stats.csv; else echo "no database for $n" > $OUT/wl_${n}_kernel_stats.csv; fi
  pass() { tag=$1; shift
    timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/wlpmc_${n}_$tag -o $tag -- python $R/tools/probes/workload.py $n 3 > $OUT/wl_${n}_pmc_$tag.log 2>&1
    db=$(find $OUT/wlpmc_${n}_$tag -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/rocpd_pmc.py "$db" > $OUT/wl_${n}_pmc_$tag.csv; fi; }
  pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU
  pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
  pass write WRITE_SIZE GRBM_GUI_ACTIVE
  rm -rf $OUT/wlprof_$n $OUT/wlpmc_${n}_sq $OUT/wlpmc_${n}_fetch $OUT/wlpmc_${n}_write $OUT/wl_${n}_prof.log $OUT/wl_${n}_pmc_*.log
  echo "== $n: $(cat $OUT/wl_${n}.json | cut -c1-200)"; head -6 $OUT/wl_${n}_kernel_stats.csv | cut -c1-150
done
cd $R
