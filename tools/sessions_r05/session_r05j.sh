#!/bin/bash
mkdir -p gpurun_out/r05j
timeout 500 python tools/probes/flake_hunt.py 7160 10 25 > gpurun_out/r05j/hunt_a.txt 2>&1; tail -14 gpurun_out/r05j/hunt_a.txt
