#!/bin/bash
# Round 6: the lean accumulate of the packed log against the templated one and the 32-bit log
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binned or bit_exact or larger_workgroups or cfg2 or attractor or pipelined or long_launch or hot" > gpurun_out/r06_third_tests.txt 2>&1
tail -3 gpurun_out/r06_third_tests.txt
L="cuburn_amd/_lib/libflame_hip_p0.so cuburn_amd/_lib/libflame_hip_p3t.so cuburn_amd/_lib/libflame_hip.so"
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for i in 1 2; do for l in $L; do
  FLAME_HIP_LIB=$PWD/$l python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l'.split('/')[-1].ljust(24), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'])"
done; done | tee gpurun_out/r06_third_ab.txt
tools/ab_prof.sh 'k_iter_spec|k_accum_tiles|k_flush' $L 2>&1 | tee gpurun_out/r06_third_abprof.txt
export FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip.so
tools/pmc_sq.sh acc_lean k_accum > gpurun_out/r06_sq_acc_lean.txt 2>&1; cat gpurun_out/r06_sq_acc_lean.txt
