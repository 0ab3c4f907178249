#!/bin/bash
mkdir -p gpurun_out/r05k
timeout 600 python tools/probes/flake_hunt2.py 7120 56 40 > gpurun_out/r05k/hunt_seq.txt 2>&1; grep -v "^pass .* done" gpurun_out/r05k/hunt_seq.txt | tail -30; tail -2 gpurun_out/r05k/hunt_seq.txt
