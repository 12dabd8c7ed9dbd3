#!/bin/bash
# A/B of two builds of the library on one box: kernels alone + frame loop, alternating.  usage: tools/r06_libab.sh <other lib> [bench args]
export TMPDIR=/tmp
mkdir -p gpurun_out
other=$PWD/$1; shift
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for rep in 1 2 3; do
  for v in shipped other; do
    if [ $v = other ]; then export FLAME_HIP_LIB=$other; else unset FLAME_HIP_LIB; fi
    echo "== $v (rep $rep)"
    tools/prof_kernels.sh libab_${v}_$rep --preheat-seconds 1.0 "$@" 2>&1 | grep -E "k_iter|k_accum" | head -2
    python bench.py --cpu-seconds 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
