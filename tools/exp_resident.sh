#!/bin/bash
# Round 4: xform records resident in scalar registers (kernels of up to FL_RESIDENT_MAX_XF xforms) against a record fetched per round
for i in 1 2; do for F in "-DFL_RESIDENT_MAX_XF=0" ""; do
  FLAME_RTC_FLAGS="$F" python3 bench.py --steps 30 --warmup 3 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 150 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; r=d['roofline']
print('flags [%s]: frame loop %.3f ms, k_iter %.3f ms per frame alone, chain %.3f ms = %.3f of 8 TB/s' % ('$F', d['ms_per_step'], k['iter'], r['chain_ms_per_frame'], r['frac']))"
done; done
