set -u
R=$PWD; OUT=$R/gpurun_out/r04u; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export BENCH_PREWARM_S=0.05 MTM_UPLOAD_BANDS=1
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extras"
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o $name -- $CMD > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed"
        db=$(find $OUT/pmc_$name -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db > $OUT/pmc_$name.csv; rm -rf $OUT/pmc_$name; }
run c1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum
run c2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
run c3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
grep -h ncc_mfma $OUT/pmc_c1.csv $OUT/pmc_c2.csv $OUT/pmc_c3.csv | sed 's/.*false>,//'
