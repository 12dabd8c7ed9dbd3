#!/bin/bash
# A/B of two library builds on one config with extra environment: tools/ab_env.sh <config> "<ENV=..>" <libA> <libB>
CFG=$1; ENVS=$2; A=$3; B=$4
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for i in 1 2; do
  for L in $A $B; do
    env $ENVS FLAME_HIP_LIB=$PWD/$L python bench.py --config $CFG --steps 6 --warmup 1 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG $ENVS', '$L'.split('/')[-1].ljust(22), d['ms_per_step'], d['kernel_ms_per_frame'])"
  done
done
