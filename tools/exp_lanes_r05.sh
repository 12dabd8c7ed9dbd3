#!/bin/bash
# round 5: stream lanes (FLAME_LANES = 2 / 3 / 4) x frames queued ahead on the frame loop
# usage: tools/exp_lanes_r05.sh [config]
cfg=${1:-cfg2}
for m in "2 2" "3 3" "4 4" "2 3" "3 2" "2 2" "3 3"; do set -- $m
  FLAME_LANES=$1 python3 bench.py --config $cfg --depth $2 --steps 40 --warmup 5 --cpu-seconds 0 --preheat-seconds 2 --min-timed-frames 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg lanes=$1 depth=$2: %.4f ms per frame, %.1f Gsamples/s' % (d['ms_per_step'], d['value']/1e3))"
done
