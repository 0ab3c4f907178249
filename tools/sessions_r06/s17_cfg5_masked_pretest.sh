# Round 6, third session: masked TM_CCORR_NORMED, hits-only - the per-output test without square root and quotient.
# (1) how cfg5's kernel time depends on the threshold (the share of the per-output normalisation), (2) A/B against the build
# before it (ab_builds/r6mid = commit 1a3f.. + nothing), alternating, (3) the masked tests + cfg5 parity tests.
set -u
TAG=${1:-r06s17}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
run() { (cd $2 && env WL_THR=$3 timeout 300 python tools/probes/workload.py cfg5 8 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1 thr=$3', d['median_ms_per_call'], d['gpu_ms'], d['ncc_kernel_ms'], d['hits'])") | tee -a $O/cfg5_ab.txt; }
for thr in 0.9 0.99 0.9999; do run mid $R/ab_builds/r6mid $thr; run new $R $thr; done
for rep in 1 2; do run mid $R/ab_builds/r6mid 0.9; run new $R 0.9; done
timeout 900 python -m pytest tests -m gpu -q -k "mask or cfg5 or last_segments" > $O/pytest_masked.log 2>&1; tail -2 $O/pytest_masked.log
