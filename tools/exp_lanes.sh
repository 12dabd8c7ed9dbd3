#!/bin/bash
# Round 4: one stream lane against two, per config (the second lane hides 0.2 ms of a cfg2 frame; at cfg3 / cfg4 / cfg5 the bench lines show it hiding nothing)
for cfg in ${@:-cfg5 cfg4 cfg3 cfg2}; do for l in 2 1 2 1; do
  FLAME_LANES=$l timeout 600 python3 bench.py --config $cfg --steps 6 --warmup 2 --cpu-seconds 0 --preheat-seconds 1.5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg FLAME_LANES=$l: frame %.3f ms  (sum of big kernels %.3f)' % (d['ms_per_step'], d['config']['stream_lanes']['sum_of_big_kernels_ms']))"
done; done
