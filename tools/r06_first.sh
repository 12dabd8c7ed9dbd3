#!/bin/bash
# Round 6, first GPU call: the packed sample log (three 21-bit records per 64-bit word) against the 32-bit log of rounds 1-5 —
# binned parity tests, frame-level and kernel-level A/B on one box.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binned or bit_exact or larger_workgroups or cfg2 or attractor or pipelined or long_launch or hot" > gpurun_out/r06_first_tests.txt 2>&1
tail -5 gpurun_out/r06_first_tests.txt
tools/ab.sh cuburn_amd/_lib/libflame_hip_p0.so cuburn_amd/_lib/libflame_hip.so 2>&1 | tee gpurun_out/r06_first_ab.txt
tools/ab_prof.sh 'k_iter_spec|k_accum_tiles|k_flush' cuburn_amd/_lib/libflame_hip_p0.so cuburn_amd/_lib/libflame_hip.so 2>&1 | tee gpurun_out/r06_first_abprof.txt
