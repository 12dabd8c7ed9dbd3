#!/bin/bash
# Round 6: directory blocks of G batches (FLAME_DIR_GROUP = log2 G) against the [tile][batch] directory of commit 3ccf514 (libflame_hip_c1.so)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "binned or bit_exact or larger_workgroups or cfg2 or attractor or pipelined or long_launch or hot or 8k or cfg5 or cfg4 or cfg3" > gpurun_out/r06_seventh_tests.txt 2>&1
tail -3 gpurun_out/r06_seventh_tests.txt
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
run() { python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1'.ljust(12), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'])"; }
for rep in 1 2; do
  FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip_c1.so run c1
  for g in 0 1 2 3; do FLAME_DIR_GROUP=$g run g$g; done
done 2>&1 | tee gpurun_out/r06_seventh_ab.txt
export FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip_c1.so; echo "== c1"; tools/prof_kernels.sh dg_c1 --preheat-seconds 1.5 | grep -E "k_iter_spec|k_accum"; unset FLAME_HIP_LIB
for g in 0 2 3; do export FLAME_DIR_GROUP=$g; echo "== g$g"; tools/prof_kernels.sh dg_$g --preheat-seconds 1.5 | grep -E "k_iter_spec|k_accum"; done
unset FLAME_DIR_GROUP
tools/pmc_traffic.sh r06b 2>&1 | head -4
