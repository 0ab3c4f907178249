# Round 6, third session: the final evidence pass once more, on the build with the division-free IEEE epilogues and the
# eight-lane device NMS (same stages as s10_final_evidence.sh; output gpurun_out/r06/, tools/collect_profiles.py r06).
set -u
R=$PWD; O=$R/gpurun_out/r06; mkdir -p $O
sed -i 's/FUZZ_N=300 bash tools\/gpu_session.sh r06 fuzz_f32/FUZZ_N=500 bash tools\/gpu_session.sh r06 fuzz_f32/' tools/session_r06_evidence.sh
bash tools/session_r06_evidence.sh > $O/evidence_stdout.log 2>&1; tail -25 $O/session.log
timeout 900 python tools/probes/f32_pieces_probe.py > $O/f32_pieces_probe.txt 2>&1; grep -c identical $O/f32_pieces_probe.txt; grep DIFFERENT $O/f32_pieces_probe.txt
timeout 600 python tools/probes/maskf32_probe.py > $O/maskf32_probe.txt 2>&1; tail -5 $O/maskf32_probe.txt
BENCH_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err; echo "2 ranks rc=$?"; cut -c1-300 $O/bench_2ranks_one_gpu.json
