#!/bin/bash
for L in libflame_hip.so libflame_hip_nw.so libflame_hip_pl.so libflame_hip_nwpl.so; do
  FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/$L FLAME_DE_CHAIN=1 python3 bench.py --steps 10 --warmup 2 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['de_filter']
print('$L chain: DE %.1f us per frame; frame loop %.3f ms' % (f['ms_per_frame']*1e3, d['ms_per_step']))"
done
