#!/bin/bash
# Same-box A/B of the shipped library against round 4's and round 5's (trees built under ab_builds/<tag> from the round-end
# commits 33b618a / ce3f2a9; untracked, they travel with the gpurun snapshot): three alternating rounds of the bench's headline
# line + map mode + dense regime, then the uint16 / cfg2 / cfg5 workloads.  Output: gpurun_out/<tag>/ab.txt
set -u
TAG=${1:-r06ab}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
clean() { grep -vE "^RCCL|^HIP|^ROCm|^Host|^Librccl" ; }
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; f=r.get('full_image_launch',{})
g=lambda k,s: (d.get(k) or {}).get(s)
print('$1', 'ms', d['ms_per_step'], 'kernel', r['kernel_ms_per_step'], 'frac', r['frac'], 'clk', r.get('sclk_mhz_in_kernel'), '| single', f.get('kernel_ms'), f.get('frac'), f.get('sclk_mhz_in_kernel'),
      '| maps', g('score_maps_materialised','ms_per_step'), g('score_maps_materialised','ncc_kernel_ms'), '| dense', g('photograph_like_image','median_ms_per_call'), g('photograph_like_image','gpu_ms'),
      '| fresh', g('fresh_templates','median_ms_per_call'), '| resident', g('resident_inputs','pipelined_ms_per_step'))"; }
for rep in 1 2 3; do
  for t in r06 r05 r04; do
    d=$R; [ $t != r06 ] && d=$R/ab_builds/$t
    (cd $d && python bench.py --no-cpu-baseline --steps 200 2>>$OUT/bench.err | clean | tail -1 | line $t) | tee -a $OUT/ab.txt
  done
done
for w in u16_4k32 cfg2 cfg5 f32_4k32; do
  for rep in 1 2; do for t in r06 r05; do
    d=$R; [ $t != r06 ] && d=$R/ab_builds/$t
    (cd $d && timeout 300 python tools/probes/workload.py $w 30 2>&1 | tail -2 | tr '\n' ' ' | sed "s/^/$t $w: /"; echo) | tee -a $OUT/ab.txt
  done; done
done
