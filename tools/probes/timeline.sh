#!/bin/bash
# rocprofv3 kernel + memory-copy timelines of matchTemplates calls (GPU box): pageable / pinned image, band layouts.
# Output: gpurun_out/<tag>/timeline_*.csv
TAG=${1:-tl}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
one() { name=$1; pin=$2; shift 2
  env "$@" python $R/tools/probes/loop_calls.py $pin 200 2>/dev/null | tail -1 | tee -a $OUT/summary.txt
  env "$@" timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/tl_$name -o tl -- python $R/tools/probes/loop_calls.py $pin 30 > $OUT/tl_$name.log 2>&1
  DB=$(find $OUT/tl_$name -name "*.db" | head -1)
  python $R/tools/rocpd_timeline.py $DB 0 100000 | tail -40 > $OUT/timeline_$name.csv 2>&1
  rm -rf $OUT/tl_$name $OUT/tl_$name.log; }
one pageable 0 MTM_X=0
one pinned 1 MTM_X=0
one pinned_b3 1 MTM_UPLOAD_BANDS=0.25,0.6,1
one pageable_b3 0 MTM_UPLOAD_BANDS=0.25,0.6,1
for b in "0.3,1" "0.2,0.55,1" "0.35,1" "0.2,1"; do MTM_UPLOAD_BANDS=$b python $R/tools/probes/loop_calls.py 0 200 2>/dev/null | tail -1 | tee -a $OUT/summary.txt; MTM_UPLOAD_BANDS=$b python $R/tools/probes/loop_calls.py 1 200 2>/dev/null | tail -1 | tee -a $OUT/summary.txt; done
