# Round 6, second session, call 4: the float32 tests and the pieces probe after the quotient-free listing screen's second
# form + 8 staging requests in flight; the two route switches that still fail, with their failure text.
set -u
O=gpurun_out/r06s4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "float32 or f32 or uint16" -s > $O/pytest_f32.log 2>&1; grep -E "passed|failed|error|float32 bound" $O/pytest_f32.log | tail -5
grep -E "^E " $O/pytest_f32.log | head -20
timeout 900 python tools/probes/f32_pieces_probe.py quick > $O/f32_pieces_probe.txt 2>&1; grep -E "cfg3_32|DIFFERENT" $O/f32_pieces_probe.txt
for e in MTM_HITS_ONLY=0 MTM_KERNEL=dot4; do
  echo "== $e"; env $e timeout 900 python -m pytest tests -m gpu -x -q > $O/alt_$e.log 2>&1; grep -E "^E|^FAILED|^ERROR|passed|failed|^tests.*Error" $O/alt_$e.log | head -30
done
