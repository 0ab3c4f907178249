set -u
O=gpurun_out/r06s23; mkdir -p $O; L=$PWD/multitemplatematching-python_amd/MTM
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; grep -E "passed|failed" $O/pytest_all.log
for rep in 1 2 3; do
  for t in new head; do
    e=""; [ $t = head ] && e="MTM_LIB_PATH=$L/libmtm_hip_head.so"
    env $e timeout 300 python tools/probes/workload.py cfg5 8 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$t cfg5', d['median_ms_per_call'], d['gpu_ms'], d['ncc_kernel_ms'], d['hits'])" | tee -a $O/masked_ab.txt
    env $e MTM_HITS_ONLY=0 timeout 300 python tools/probes/workload.py cfg5 6 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$t cfg5 maps in memory', d['median_ms_per_call'], d['gpu_ms'], d['ncc_kernel_ms'], d['hits'], d['hits_only'])" | tee -a $O/masked_ab.txt
  done
done
timeout 600 python tools/fuzz_parity.py 7000 200 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
