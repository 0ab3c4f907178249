# Round 6, third session: quotient_as_float (the IEEE-division epilogues without the division sequence).
# (1) the function against the division (2^31 cases) + the whole-map last-segment tests, (2) the GPU suite,
# (3) same-box A/B against the build before it (ab_builds/r6base = commit 0f0c0d1's tree), (4) MTM_KERNEL=dot4 over the suite.
set -u
TAG=${1:-r06s13}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
clean() { grep -vE "^RCCL|^HIP|^ROCm|^Host|^Librccl" ; }
timeout 600 python -m pytest tests/test_gpu_last_segments.py -q -x > $O/pytest_last_segments.log 2>&1; tail -3 $O/pytest_last_segments.log
python - <<'PY' 2>&1 | tee $O/quotient_check.txt
import sys; sys.path.insert(0, "multitemplatematching-python_amd")
from MTM import _lib
c = _lib.Context(0)
import time
for seed in (11, 12):
    t = time.time(); r = c.debug_quotient_check(1 << 30, seed); print(seed, r, "%.2f s" % (time.time() - t))
PY
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; grep -E "passed|failed" $O/pytest_all.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; f=r.get('full_image_launch',{})
g=lambda k,s: (d.get(k) or {}).get(s)
print('$1', 'ms', d['ms_per_step'], 'kernel', r['kernel_ms_per_step'], 'frac', r['frac'], 'clk', r.get('sclk_mhz_in_kernel'), '| single', f.get('kernel_ms'), f.get('frac'),
      '| maps', g('score_maps_materialised','ms_per_step'), g('score_maps_materialised','ncc_kernel_ms'), '| dense', g('photograph_like_image','median_ms_per_call'), g('photograph_like_image','gpu_ms'),
      '| fresh', g('fresh_templates','median_ms_per_call'), '| resident', g('resident_inputs','pipelined_ms_per_step'))"; }
for rep in 1 2 3; do
  for t in new base; do
    d=$R; [ $t != new ] && d=$R/ab_builds/${AB_BASE:-r6base}
    (cd $d && python bench.py --no-cpu-baseline --steps 200 2>>$O/bench.err | clean | tail -1 | line $t) | tee -a $O/ab.txt
  done
done
true
