# Round 6, evidence pass (profiles/README.md): suite + driver-flag bench + rocprofv3 kernel trace / PMC of the headline and the
# secondary workloads + group runs, then the suite in two more orders, then the fuzzers (uint8 / uint16 seeds 7000-7500 - the
# range of round 5's unreproduced difference - and 300 float32 cases).  One gpurun call; tools/session_r06_alt.sh is the other.
set -u
export PROFILE_TAG=r06 TL_CASES="default" TL_TAIL=30
bash tools/gpu_session.sh r06 tests_all bench_driver profile workloads group
for order in reverse shuffle:6; do
  MTM_TEST_ORDER=$order timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r06/pytest_order_${order/:/_}.log 2>&1
  echo "order $order: $(tail -1 gpurun_out/r06/pytest_order_${order/:/_}.log)" | tee -a gpurun_out/r06/session.log
done
timeout 900 python tools/fuzz_parity.py 7000 500 > gpurun_out/r06/fuzz_7000_7500.txt 2>&1; tail -2 gpurun_out/r06/fuzz_7000_7500.txt | tee -a gpurun_out/r06/session.log
FUZZ_N=300 bash tools/gpu_session.sh r06 fuzz_f32
python tools/probes/dense_probe.py 0.5 2>/dev/null | grep -E "call median|dense call" > gpurun_out/r06/dense.txt; cat gpurun_out/r06/dense.txt
