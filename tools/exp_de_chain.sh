#!/bin/bash
# Round 4: the DE as one persistent launch (FLAME_DE_CHAIN=1) against one kernel per direction (0) and against the
# persistent launch's tile shapes run one direction per launch (2): DE time per frame (HIP events, one lane) and the
# frame loop.  usage: tools/exp_de_chain.sh [config ...]
for cfg in ${@:-cfg2}; do for m in 0 1 2 1 0; do
  FLAME_DE_CHAIN=$m python3 bench.py --config $cfg --steps 20 --warmup 3 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 60 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['de_filter']
print('$cfg FLAME_DE_CHAIN=$m: DE %.1f us per frame = %.3f of the copy rate (%.0f GB/s), %.3f of 6.3 TB/s; frame loop %.3f ms' % (f['ms_per_frame']*1e3, f['frac_of_copy'], f['measured_copy_gbps'], f['frac_of_achievable_6300'], d['ms_per_step']))"
done; done
