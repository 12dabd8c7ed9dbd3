#!/bin/bash
# A/B of two builds of the library on the same box: tools/ab.sh <libA> <libB> [bench args]
A=$1; B=$2; shift 2
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for i in 1 2; do
  for L in $A $B; do
    FLAME_HIP_LIB=$PWD/$L python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1].ljust(24), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'])"
  done
done
