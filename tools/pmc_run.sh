#!/bin/bash
# PMC passes for the dominant kernel (run on the GPU box from the repo root).  Each pass is its own
# rocprofv3 run with --kernel-trace only (no sys/hip/hsa tracing), as the pool requires.
set -u
R=$PWD; OUT=$R/gpurun_out/pmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU
run p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run p3 FETCH_SIZE GRBM_GUI_ACTIVE
run p4 WRITE_SIZE GRBM_GUI_ACTIVE
run p5 TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
cd $R
