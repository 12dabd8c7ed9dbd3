#!/bin/bash
# Round 4: wave-uniform operands of the round (affine / camera offsets, colour products) held in vector registers for the launch
# (FL_HOIST_BUDGET, iter.hip) against rebuilt every round (FLAME_RTC_FLAGS=-DFL_HOIST_BUDGET=0).  usage: tools/exp_hoist.sh [config ...]
for cfg in ${@:-cfg2 cfg3 cfg5}; do for f in "" "-DFL_HOIST_BUDGET=0" "" "-DFL_HOIST_BUDGET=0"; do
  FLAME_RTC_FLAGS="$f" timeout 600 python3 bench.py --config $cfg --steps 10 --warmup 2 --cpu-seconds 0 --preheat-seconds 1.5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; r=d['roofline']
print('$cfg [%-20s] frame %.3f ms  iterate alone %.3f  chain frac %.3f' % ('$f', d['ms_per_step'], k.get('iter', 0), r['frac']))"
done; done
