#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_fifth_tests.txt 2>&1
tail -5 gpurun_out/r06_fifth_tests.txt
tools/ab.sh cuburn_amd/_lib/libflame_hip_p0.so cuburn_amd/_lib/libflame_hip.so 2>&1 | grep '^libflame' | tee gpurun_out/r06_fifth_ab.txt
