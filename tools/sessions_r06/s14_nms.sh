# Round 6, third session: the device NMS's neighbourhood passes with eight lanes per candidate.
# (1) the tests that reach the device NMS, (2) kernel trace of the dense workload, new against ab_builds/r6base,
# (3) the dense probe / bench line A/B, alternating.
set -u
TAG=${1:-r06s14}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
clean() { grep -vE "^RCCL|^HIP|^ROCm|^Host|^Librccl" ; }
timeout 900 python -m pytest tests -m gpu -q -k "nms or dense or photograph or border or peaks" > $O/pytest_nms.log 2>&1; tail -3 $O/pytest_nms.log
MTM_NMS_DEVICE_MIN=-1 timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_device_nms_always.log 2>&1; tail -2 $O/pytest_device_nms_always.log
for t in new base; do
  d=$R; [ $t != new ] && d=$R/ab_builds/r6base
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$t -o prof -- python $d/tools/probes/workload.py dense_4k32_nms 6 > $O/prof_$t.log 2>&1)
  DB=$(find $O/prof_$t -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py "$DB" > $O/dense_nms_kernel_stats_$t.csv
  rm -rf $O/prof_$t
  echo "== $t"; cut -c1-60,200-260 $O/dense_nms_kernel_stats_$t.csv | head -14
done
for rep in 1 2 3; do for t in new base; do
  d=$R; [ $t != new ] && d=$R/ab_builds/r6base
  (cd $d && timeout 300 python tools/probes/workload.py dense_4k32_nms 40 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['median_ms_per_call'], d['gpu_ms'], d['ncc_kernel_ms'], d['hits'])") | tee -a $O/dense_ab.txt
done; done
