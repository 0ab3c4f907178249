#!/bin/bash
mkdir -p gpurun_out/r05m
FUZZ_DTYPE=uint16 FLAKE_POLLUTE=0 timeout 400 python tools/probes/flake_hunt2.py 7000 200 12 > gpurun_out/r05m/hunt_u16.txt 2>&1; grep -v "^pass .* done" gpurun_out/r05m/hunt_u16.txt | tail -12
FLAKE_POLLUTE=0 timeout 400 python tools/probes/flake_hunt2.py 7000 200 8 > gpurun_out/r05m/hunt_mixed.txt 2>&1; grep -v "^pass .* done" gpurun_out/r05m/hunt_mixed.txt | tail -12
