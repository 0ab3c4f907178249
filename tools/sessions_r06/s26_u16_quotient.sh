# Round 6, last session: the division-free quotient in the uint16 epilogue - suite, uint16 maps in memory A/B against the build before.
set -u
O=gpurun_out/r06s26; mkdir -p $O; L=$PWD/multitemplatematching-python_amd/MTM
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; grep -E "passed|failed" $O/pytest_all.log
for rep in 1 2 3; do for t in new head; do
  e=""; [ $t = head ] && e="MTM_LIB_PATH=$L/libmtm_hip_head.so"
  for ho in 0 1; do
    env $e MTM_HITS_ONLY=$ho timeout 300 python tools/probes/workload.py u16_4k32 12 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$t u16 hits_only=$ho', d['median_ms_per_call'], d['gpu_ms'], d['ncc_kernel_ms'], d['hits'], d['hits_only'])" | tee -a $O/u16_ab.txt
  done
done; done
FUZZ_DTYPE=uint16 timeout 600 python tools/fuzz_parity.py 30000 300 > $O/fuzz_u16.txt 2>&1; tail -2 $O/fuzz_u16.txt
