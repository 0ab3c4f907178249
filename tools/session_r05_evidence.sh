# Round 5's evidence pass: one gpurun call, one box (profiles/README.md)
set -u
export PROFILE_TAG=r05 TL_CASES="default X=0" TL_TAIL=30
bash tools/gpu_session.sh r05 tests bench_driver profile workloads group trace timeline
for i in 1 2 3; do
  python tools/probes/rccl_first_probe.py 200 2>/dev/null | grep "communicator first" | tee -a gpurun_out/r05/rccl_first.txt
  python tools/probes/group_calls.py 1 200 2>/dev/null | grep "group of" | tee -a gpurun_out/r05/group_calls.txt
  python tools/probes/group_calls.py 0 200 2>/dev/null | grep "group of" | tee -a gpurun_out/r05/group_calls.txt
done
