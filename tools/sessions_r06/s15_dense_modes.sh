set -u
O=gpurun_out/r06s15; mkdir -p $O
for rep in 1 2; do for e in "" "MTM_SEG_SKIP=0" "MTM_SPARSE_MAPS=0"; do
  env $e timeout 300 python tools/probes/workload.py dense_4k32_nms 40 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('[$e]', d['median_ms_per_call'], d['gpu_ms'], d['ncc_kernel_ms'], d['hits'], d['hits_only'])" | tee -a $O/dense_modes.txt
done; done
MTM_HOST_TRACE=1 python tools/probes/workload.py dense_4k32_nms 3 2>&1 | tail -40 > $O/host_trace.txt
