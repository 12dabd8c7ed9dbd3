#!/bin/bash
# Round 6: k_accum_tiles<8> at 8K with its partial wait back inside the first-iteration test, against round 5's library (libflame_hip_r05.so)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "binned or attractor or long_launch or 8k or cfg5 or cfg4 or pipelined" > gpurun_out/r06_waits_tests.txt 2>&1
tail -3 gpurun_out/r06_waits_tests.txt
for cfg in cfg5 cfg2; do for L in "" _r05 "" _r05; do
  export FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip$L.so
  echo "== $cfg lib$L"; tools/prof_kernels.sh w$cfg$L --config $cfg --min-timed-frames 24 | grep -E "k_accum"
done; done 2>&1 | tee gpurun_out/r06_waits.txt
