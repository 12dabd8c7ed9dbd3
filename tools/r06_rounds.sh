#!/bin/bash
# Rounds per sorted batch above 16 (libraries built with -DFL_BIN_R_MAX=24 / 28): kernels alone + frame loop, one box.
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for rep in 1 2; do
  for r in 16 20; do
    if [ $r = 16 ]; then unset FLAME_HIP_LIB FLAME_BIN_ROUNDS; else export FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip_r$r.so FLAME_BIN_ROUNDS=$r; fi
    echo "== rounds $r (rep $rep)"
    tools/prof_kernels.sh rounds_${r}_$rep --preheat-seconds 1.0 $BENCH_ARGS 2>&1 | grep -E "k_iter|k_accum" | head -2
    python bench.py --cpu-seconds 0 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
