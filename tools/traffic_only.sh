#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the bench command only (hits-only default), into gpurun_out/<tag>/
TAG=${1:-traffic}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp BENCH_PREWARM=4; cd /tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extras"
for spec in "p3 FETCH_SIZE" "p4 WRITE_SIZE"; do set -- $spec
  timeout 600 rocprofv3 --kernel-trace --pmc $2 GRBM_GUI_ACTIVE -d $OUT/pmc_$1 -o $1 -- $CMD > $OUT/pmc_$1.log 2>&1
  db=$(find $OUT/pmc_$1 -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db > $OUT/pmc_$1.csv; rm -rf $OUT/pmc_$1
done
cd $R; grep -h ncc_mfma $OUT/pmc_p3.csv $OUT/pmc_p4.csv | grep -E "FETCH|WRITE"
