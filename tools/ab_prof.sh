#!/bin/bash
# Precise kernel-level A/B on one box: rocprofv3 average durations over ~600 launches per library.
# usage: tools/ab_prof.sh <kernel-regex> <libA> <libB> [...]
pat=$1; shift
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for rep in 1 2; do for L in "$@"; do
  export FLAME_HIP_LIB=$PWD/$L
  echo "== $L"; tools/prof_kernels.sh ab_$(basename $L .so)_$rep --preheat-seconds 1.5 | grep -E "$pat"
done; done
# which kind of box was this?  (some boxes of the pool run the ALU-bound kernels ~1.9x slower throughout)
rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -E "sclk|mclk|Power|Performance" | head -6
