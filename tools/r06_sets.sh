#!/bin/bash
# Round 6: register sets of k_accum_tiles_p3's record loop (ACC_P3_SETS - 1 steps of 64 words in flight): 3, 4, 5
mkdir -p gpurun_out
L="cuburn_amd/_lib/libflame_hip_s3.so cuburn_amd/_lib/libflame_hip_s4.so cuburn_amd/_lib/libflame_hip_s5.so"
for l in $L; do FLAME_HIP_LIB=$PWD/$l timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binned or attractor" 2>&1 | tail -1; done
tools/ab_prof.sh 'k_accum_tiles' $L 2>&1 | grep -v '^GPU\|^=====' | tee gpurun_out/r06_sets.txt
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for i in 1 2; do for l in $L; do
  FLAME_HIP_LIB=$PWD/$l python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l'.split('/')[-1].ljust(24), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'])"
done; done | tee -a gpurun_out/r06_sets.txt
for l in $L; do export FLAME_HIP_LIB=$PWD/$l; echo "== cfg4 $l"; tools/prof_kernels.sh s4 --config cfg4 --min-timed-frames 24 | grep -E "k_accum"; done | tee -a gpurun_out/r06_sets.txt
