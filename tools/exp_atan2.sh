#!/bin/bash
# Round 4: the lean atan2 of variations.h against the device library's atan2f (FLAME_RTC_FLAGS=-DFL_LIBM_MATH compiles the per-genome
# kernel with the latter): bench lines of the configs whose flames use it.  usage: tools/exp_atan2.sh [config ...]
for cfg in ${@:-cfg5 cfg3 cfg4 cfg2}; do for f in "" "-DFL_LIBM_MATH" "" "-DFL_LIBM_MATH"; do
  FLAME_RTC_FLAGS="$f" timeout 600 python3 bench.py --config $cfg --steps 6 --warmup 2 --cpu-seconds 0 --preheat-seconds 1.5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; r=d['roofline']
print('$cfg [%-16s] frame %.3f ms  iterate %.3f  accumulate %.3f  chain frac %.3f' % ('$f', d['ms_per_step'], k.get('iterate', 0), k.get('accumulate', 0), r['frac']))"
done; done
