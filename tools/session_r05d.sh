set -u
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -1 $O/pytest.log | grep -E "passed|failed" ; grep -E "passed|failed" $O/pytest.log | tail -1
python tools/probes/rccl_probe.py 200 2>/dev/null | grep -E "context|again" | tee $O/rccl_probe.txt
for i in 1 2; do
  python tools/probes/group_calls.py 0 200 2>/dev/null | grep "group of" | tee -a $O/group.txt
  python tools/probes/group_calls.py 1 200 2>/dev/null | grep "group of" | tee -a $O/group.txt
done
