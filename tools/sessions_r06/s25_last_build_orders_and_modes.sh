# Round 6, last session: the suite on the very last build (after the RGB and masked forms of the division-free quotient) in two
# more orders and under the route switches that reach those epilogues.
set -u
O=gpurun_out/r06s25; mkdir -p $O
for order in reverse shuffle:6; do
  MTM_TEST_ORDER=$order timeout 900 python -m pytest tests -m gpu -q > $O/pytest_order_${order/:/_}.log 2>&1
  echo "order $order: $(grep -E 'passed|failed' $O/pytest_order_${order/:/_}.log)" | tee -a $O/summary.txt
done
ALT_MODES="MTM_ROW_MUX=0 MTM_HITS_ONLY=0 MTM_EXACT_DIV=0 MTM_EXACT_DIV=2 MTM_MASKSQ_FUSED=0 MTM_KERNEL=dot4" bash tools/alt_modes.sh 2>&1 | tee -a $O/summary.txt
