#!/bin/bash
# frame loop with the DE's launches queued per fl_filter call (FLAME_DE_CHAIN=0), all at the end (3), or as the
# persistent launch's tile shapes one direction per launch (2)
for i in 1 2 3; do for m in 0 3 2; do
  FLAME_DE_CHAIN=$m python3 bench.py --steps 30 --warmup 3 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 150 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['de_filter']
print('FLAME_DE_CHAIN=$m: frame loop %.3f ms  (DE alone %.1f us, fuse-64 loop %.3f ms)' % (d['ms_per_step'], f['ms_per_frame']*1e3, d['config']['fuse_short']['ms_per_step']))"
done; done
