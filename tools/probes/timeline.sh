#!/bin/bash
# rocprofv3 kernel + memory-copy timelines of matchTemplates calls (GPU box).
# Usage: timeline.sh <tag>          cases from TL_CASES="name env=.. env=..;name2 env=.." (default: the shipped layout and
#                                   the first band on the copy stream); pageable image, 30 calls traced, 200 timed
#                                   TL_CMD="tools/probes/workload.py slab_414": the command traced instead of loop_calls.py
# Output: gpurun_out/<tag>/timeline_<name>.csv (the last 40 GPU activities) + summary.txt (per-call medians)
TAG=${1:-tl}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD=${TL_CMD:-tools/probes/loop_calls.py 0}
one() { name=$1; shift
  env "$@" python $R/$CMD 200 2>/dev/null | tail -1 | cut -c1-300 | sed "s/^/$name: /" | tee -a $OUT/summary.txt
  env "$@" timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/tl_$name -o tl -- python $R/$CMD 30 > $OUT/tl_$name.log 2>&1
  DB=$(find $OUT/tl_$name -name "*.db" | head -1)
  python $R/tools/rocpd_timeline.py $DB 0 100000 | tail -${TL_TAIL:-40} > $OUT/timeline_$name.csv 2>&1
  rm -rf $OUT/tl_$name $OUT/tl_$name.log; }
IFS=';' read -ra CASES <<< "${TL_CASES:-inline MTM_BAND_INLINE=1;copy_stream MTM_BAND_INLINE=0}"
for cs in "${CASES[@]}"; do one $cs; done
