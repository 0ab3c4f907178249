#!/bin/bash
# Sample the engine clock / power while the bench workload keeps the GPU busy (GPU box).
python bench.py --steps 15000 --warmup 5 --no-cpu-baseline > /tmp/bench_bg.log 2>&1 &
BG=$!
for i in $(seq 1 60); do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power|GPU use" | tr '\n' ' ' | sed 's/=*//g'
  echo
  kill -0 $BG 2>/dev/null || break
  sleep 1
done
wait $BG
tail -1 /tmp/bench_bg.log | cut -c1-200
