#!/bin/bash
# A/B of two builds of the library on another BASELINE config: tools/ab_cfg.sh <config> <libA> <libB>
CFG=$1; A=$2; B=$3
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for i in 1 2; do
  for L in $A $B; do
    FLAME_HIP_LIB=$PWD/$L python bench.py --config $CFG --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1].ljust(24), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'])"
  done
done
