#!/bin/bash
# Round 6 against round 5's library (commit b9ebaec built as cuburn_amd/_lib/libflame_hip_r05.so) on ONE box: frame loop and kernels, every config
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 3 > /dev/null 2>&1
run() { python bench.py --config $2 --steps $3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1 --min-timed-frames $4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2 $1'.ljust(12), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'], d['roofline']['frac'])"; }
for cfg in cfg2 cfg3 cfg4 cfg5; do
  case $cfg in cfg2) st=8; mf=300;; cfg3) st=6; mf=100;; cfg4) st=6; mf=120;; cfg5) st=4; mf=16;; esac
  for rep in 1 2; do
    FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip_r05.so run r05 $cfg $st $mf
    run r06 $cfg $st $mf
  done
done 2>&1 | tee gpurun_out/r06_vs_r05.txt
for L in _r05 ""; do
  export FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip$L.so
  echo "== cfg5 lib$L"; tools/prof_kernels.sh v5$L --config cfg5 --min-timed-frames 16 | grep -E "k_iter_spec|k_accum"
done 2>&1 | tee -a gpurun_out/r06_vs_r05.txt
