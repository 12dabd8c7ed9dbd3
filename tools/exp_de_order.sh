#!/bin/bash
for i in 1 2; do for m in "0 x" "2 x" "0 0"; do set -- $m
  if [ $2 = x ]; then unset FLAME_DE_ORDER; else export FLAME_DE_ORDER=$2; fi
  FLAME_DE_CHAIN=$1 python3 bench.py --steps 30 --warmup 3 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 150 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['de_filter']
print('FLAME_DE_CHAIN=$1 FLAME_DE_ORDER=$2: frame loop %.3f ms  (DE alone %.1f us, fuse-64 loop %.3f ms)' % (d['ms_per_step'], f['ms_per_frame']*1e3, d['config']['fuse_short']['ms_per_step']))"
done; done
