# Round 6, second session, call 6: the phase stagger of ncc_bf16_kernel's work-groups, on / off, one box, alternating.
set -u
O=gpurun_out/r06s6; mkdir -p $O
for rep in 1 2; do for st in 1 0; do
  echo "== MTM_BF16_STAGGER=$st (round $rep)"; MTM_BF16_STAGGER=$st timeout 600 python tools/probes/f32_pieces_probe.py quick 2>&1 | grep -E "cfg3_32 sparse image, (TM_CCOEFF_NORMED thr|masked TM_CCORR_NORMED thr|TM_CCOEFF_NORMED N_object)|DIFFERENT"
done; done 2>&1 | tee $O/stagger_ab.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "float32 or f32 or uint16" > $O/pytest_f32.log 2>&1; grep -E "passed|failed|error" $O/pytest_f32.log | tail -3
