#!/bin/bash
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg5 -o p -- python $R/bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_cfg5.log 2>&1
DB=$(find $R/gpurun_out/prof_cfg5 -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB | cut -c1-200 | head -14
rm -rf $R/gpurun_out/prof_cfg5
