# Round 6, second session, call 3: rocprofv3 of the float32 workload with the one-product screen (kernel trace + SQ counters +
# wait states), bench.py as two ranks on ONE GPU (RCCL refuses a shared device: every rank must take the gloo fall-back
# together), and the suite under the four route switches that failed or are new.
set -u
R=$PWD; O=$R/gpurun_out/r06s3; mkdir -p $O
export TMPDIR=/tmp
bash tools/profile_workloads.sh r06s3 f32_4k32 > $O/profile_f32.log 2>&1; tail -8 $O/profile_f32.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $O/pmc_wait -o wait -- python $R/tools/probes/workload.py f32_4k32 3 > $O/pmc_wait.log 2>&1
db=$(find $O/pmc_wait -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_pmc.py "$db" > $O/wl_f32_4k32_pmc_wait.csv; rm -rf $O/pmc_wait
cd $R
grep -h bf16 $O/wl_f32_4k32_pmc_sq.csv $O/wl_f32_4k32_pmc_wait.csv | cut -c1-400
echo "== two ranks on one GPU"
BENCH_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err; echo "rc=$?"; cut -c1-700 $O/bench_2ranks_one_gpu.json; grep -i "fall\|error\|Traceback" $O/bench_2ranks_one_gpu.err | head -5
ALT_MODES="MTM_F32_MFMA=3 MTM_HITS_ONLY=0 MTM_KERNEL=dot4 MTM_F32_MFMA=2" bash tools/alt_modes.sh > $O/alt_modes_subset.txt 2>&1; cat $O/alt_modes_subset.txt
