# Round 6, second session, call 8: float32 images in two upload bands - the fused-call test (banded == unbanded, bit for bit),
# the float32 tests, the pieces probe with and without bands (MTM_UPLOAD_BANDS=1: one piece), then the whole suite.
set -u
O=gpurun_out/r06s8; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "fused_image_call or float32 or f32" > $O/pytest_bands.log 2>&1; grep -E "passed|failed|error" $O/pytest_bands.log | tail -3; grep -E "^E " $O/pytest_bands.log | head -20
for rep in 1 2; do for b in "0.25,1" "1"; do
  echo "== MTM_UPLOAD_BANDS=$b (round $rep)"; MTM_UPLOAD_BANDS=$b timeout 600 python tools/probes/f32_pieces_probe.py quick 2>&1 | grep -E "cfg3_32 sparse image, (TM_CCOEFF_NORMED thr|TM_CCOEFF_NORMED N_object|TM_CCOEFF \(raw\) thr)|DIFFERENT"
done; done 2>&1 | tee $O/bands_ab.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; grep -E "passed|failed" $O/pytest_all.log; grep -E "^E |^FAILED" $O/pytest_all.log | head
FUZZ_N=300 bash tools/gpu_session.sh r06s8 fuzz_f32 | tail -2
