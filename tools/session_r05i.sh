set -u
O=$PWD/gpurun_out/r05i; mkdir -p $O; R=$PWD
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o prof -- python $R/bench.py --config cfg5 --no-cpu-baseline --skip-extras --steps 5 --warmup 2 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/kernel_stats_cfg5.csv
rm -rf $O/prof
head -14 $O/kernel_stats_cfg5.csv | cut -c1-200
