#!/bin/bash
mkdir -p gpurun_out/r05i
{
for start in 7168 7167 7165 7160 7150 7130 7100 7000; do
  n=$((7170-start))
  echo "== from $start ($n cases)"; timeout 300 python tools/fuzz_parity.py $start $n 2>&1 | tail -3
done
echo "== again from 7000"; timeout 300 python tools/fuzz_parity.py 7000 170 2>&1 | tail -3
} > gpurun_out/r05i/prefix.txt 2>&1
cat gpurun_out/r05i/prefix.txt
