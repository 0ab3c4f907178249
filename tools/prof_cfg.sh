#!/bin/bash
# rocprofv3 kernel summary of bench.py --config <cfg>
CFG=${1:-cfg2}; export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$CFG -o p -- python $R/bench.py --config $CFG --steps 50 --warmup 5 --no-cpu-baseline --skip-extras > $R/gpurun_out/prof_$CFG.log 2>&1
DB=$(find $R/gpurun_out/prof_$CFG -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB | cut -c1-60,180-260 | head -10
tail -1 $R/gpurun_out/prof_$CFG.log | cut -c1-150
rm -rf $R/gpurun_out/prof_$CFG
