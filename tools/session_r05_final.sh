# Round 5, final build: the evidence pass (profiles/README.md) + fuzzers + the round's last switches.  One gpurun call.
set -u
export PROFILE_TAG=r05 TL_CASES="default" TL_TAIL=30
bash tools/gpu_session.sh r05 tests bench_driver profile workloads group
FUZZ_N=300 bash tools/gpu_session.sh r05 fuzz_f32
timeout 400 python tools/fuzz_parity.py 0 300 > gpurun_out/r05/fuzz_all.txt 2>&1; tail -2 gpurun_out/r05/fuzz_all.txt
ALT_MODES="${FINAL_ALT:-MTM_SEG_SKIP=0 MTM_F32_RIG=0 MTM_HITS_ONLY=0}" bash tools/alt_modes.sh > gpurun_out/r05/alt_modes.txt 2>&1; cat gpurun_out/r05/alt_modes.txt
python tools/probes/f32_raw_probe.py 2>/dev/null | grep cfg > gpurun_out/r05/f32_raw.txt; cat gpurun_out/r05/f32_raw.txt
python tools/probes/dense_probe.py 0.5 2>/dev/null | grep -E "call median|dense call" > gpurun_out/r05/dense.txt; cat gpurun_out/r05/dense.txt
