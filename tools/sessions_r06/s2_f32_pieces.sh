# Round 6, second session, call 2: the one-product tier of the float32 routes - its tests, the float32 tests around it, the
# probe (per-call times against the three-product screen and the float64 kernel), then the whole suite.
set -u
O=gpurun_out/r06s2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "float32 or f32 or uint16" -s > $O/pytest_f32.log 2>&1; grep -E "passed|failed|error|one-product tier|float32 bound" $O/pytest_f32.log | tail -8
grep -E "^E " $O/pytest_f32.log | head -20
timeout 900 python tools/probes/f32_pieces_probe.py > $O/f32_pieces_probe.txt 2>&1; cat $O/f32_pieces_probe.txt | tail -70
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; grep -E "passed|failed" $O/pytest_all.log
