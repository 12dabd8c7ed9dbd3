#!/bin/bash
# Round 4: the DE's eight launches overlapped on two streams (FLAME_DE_CHAIN=4: direction p + 1 starts under the tail of p,
# de_chain.hip k_de_lap) against the default (3: eight launches back to back) and the same tiles one per launch (2):
# DE time per frame (HIP events, one lane) and the two-lane frame loop.  usage: tools/exp_de_lap.sh [config ...]
for cfg in ${@:-cfg2}; do for m in 3 4 2 4 3 4; do
  FLAME_DE_CHAIN=$m timeout 300 python3 bench.py --config $cfg --steps 20 --warmup 3 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 60 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['de_filter']
print('$cfg FLAME_DE_CHAIN=$m: DE %.1f us per frame = %.3f of the copy rate (%.0f GB/s), %.3f of 6.3 TB/s; frame loop %.3f ms' % (f['ms_per_frame']*1e3, f['frac_of_copy'], f['measured_copy_gbps'], f['frac_of_achievable_6300'], d['ms_per_step']))"
done; done
