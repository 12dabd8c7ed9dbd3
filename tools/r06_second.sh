#!/bin/bash
# Round 6: packed log A/B with the per-genome kernel, + SQ counters of the accumulate under both formats
mkdir -p gpurun_out
tools/ab.sh cuburn_amd/_lib/libflame_hip_p0.so cuburn_amd/_lib/libflame_hip.so 2>&1 | grep '^libflame' | tee gpurun_out/r06_second_ab.txt
tools/ab_prof.sh 'k_iter_spec|k_accum_tiles|k_flush' cuburn_amd/_lib/libflame_hip_p0.so cuburn_amd/_lib/libflame_hip.so 2>&1 | tee gpurun_out/r06_second_abprof.txt
for L in p0 p3; do
  if [ $L = p0 ]; then export FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip_p0.so; else export FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip.so; fi
  tools/pmc_sq.sh acc_$L k_accum > gpurun_out/r06_sq_acc_$L.txt 2>&1
  tools/pmc_sq.sh it_$L k_iter_spec > gpurun_out/r06_sq_it_$L.txt 2>&1
done
paste gpurun_out/r06_sq_acc_p0.txt gpurun_out/r06_sq_acc_p3.txt
paste gpurun_out/r06_sq_it_p0.txt gpurun_out/r06_sq_it_p3.txt
