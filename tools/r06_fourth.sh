#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binned or attractor" > gpurun_out/r06_fourth_tests.txt 2>&1
tail -3 gpurun_out/r06_fourth_tests.txt
L="cuburn_amd/_lib/libflame_hip_p0.so cuburn_amd/_lib/libflame_hip_p3t.so cuburn_amd/_lib/libflame_hip.so"
tools/ab_prof.sh 'k_iter_spec|k_accum_tiles|k_flush' $L 2>&1 | grep -v 'k_iter_spec\|k_flush' | tee gpurun_out/r06_fourth_abprof.txt
export FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip.so
tools/pmc_sq.sh acc_lean2 k_accum > gpurun_out/r06_sq_acc_lean2.txt 2>&1; cat gpurun_out/r06_sq_acc_lean2.txt
