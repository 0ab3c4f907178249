# Round 6, second session, last call: bench.py with the float32 extra (driver's flags and the default), smoke(), and the
# GPU suite once more on exactly what is committed.
set -u
O=gpurun_out/r06s12; mkdir -p $O
clean() { grep -vE "^RCCL|^HIP|^ROCm|^Host|^Librccl" ; }
python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | clean | tail -1 > $O/bench_driver.json
python -c "
import json; d=json.load(open('$O/bench_driver.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print(json.dumps(d.get('float32_image'), indent=1))"
python bench.py 2>>$O/bench.err | clean | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['fresh_templates']['median_ms_per_call'], d['score_maps_materialised']['ncc_kernel_ms'], d['photograph_like_image']['median_ms_per_call'], d['float32_image'])"
python __graft_entry__.py 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; grep -E "passed|failed" $O/pytest_all.log
echo "== MTM_KERNEL=dot4"; MTM_KERNEL=dot4 timeout 900 python -m pytest tests -m gpu -x -q > $O/alt_dot4.log 2>&1; grep -E "^E|^FAILED|^ERROR|passed|failed|^tests.*Error|^tests/.*py:[0-9]+" $O/alt_dot4.log | head -30
