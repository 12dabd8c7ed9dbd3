#!/bin/bash
# Round 4: the per-xform operand table in LDS of per-genome kernels whose records are fetched per round (kTab, FL_HOIST_BUDGET >= 12) against
# the same kernels without it (-DFL_HOIST_BUDGET=7).  usage: tools/exp_tab.sh [config ...]
for cfg in ${@:-cfg3 cfg4 cfg5}; do for f in "" "-DFL_HOIST_BUDGET=7" "" "-DFL_HOIST_BUDGET=7"; do
  FLAME_RTC_FLAGS="$f" timeout 600 python3 bench.py --config $cfg --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1.5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; r=d['roofline']
print('$cfg [%-20s] frame %.3f ms  iterate alone %.3f  chain frac %.3f' % ('$f', d['ms_per_step'], k.get('iter', 0), r['frac']))"
done; done
