#!/bin/bash
mkdir -p gpurun_out/r05s2
timeout 700 python tools/probes/bisect_suite.py test_uint16_many_templates > gpurun_out/r05s2/bisect.txt 2>&1
cat gpurun_out/r05s2/bisect.txt | cut -c1-250
