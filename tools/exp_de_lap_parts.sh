#!/bin/bash
# Round 4: where the overlapped DE launches (FLAME_DE_CHAIN=4) lose their time: FLAME_DE_LAP_DBG bits knock parts out
# (1 no waits, 2 no publishing, 4 one stream / no gates, 8 no started count, 16 plain loads and stores).  Results are wrong.
for dbg in ${DBGS:-0 4 1 2 3 16 0}; do
  FLAME_DE_LAP_DBG=$dbg FLAME_DE_CHAIN=4 timeout 300 python3 bench.py --steps 10 --warmup 2 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['de_filter']
print('FLAME_DE_LAP_DBG=$dbg: DE %.1f us per frame; frame loop %.3f ms' % (f['ms_per_frame']*1e3, d['ms_per_step']))"
done
