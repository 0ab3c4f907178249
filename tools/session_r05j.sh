set -u
O=$PWD/gpurun_out/r05j; mkdir -p $O; R=$PWD
export TMPDIR=/tmp; cd /tmp
for v in "" nocomp noload; do
  lib=$R/multitemplatematching-python_amd/MTM/libmtm_hip${v:+_$v}.so
  MTM_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof$v -o prof -- python $R/bench.py --config cfg5 --no-cpu-baseline --skip-extras --steps 5 --warmup 2 > $O/prof$v.log 2>&1
  DB=$(find $O/prof$v -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB > $O/kernel_stats_cfg5_$v.csv
  rm -rf $O/prof$v
  echo "== ${v:-product}"; grep "masksq_runs" $O/kernel_stats_cfg5_$v.csv | cut -d, -f2-7
done
