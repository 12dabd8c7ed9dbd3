#!/bin/bash
# Round 4: what the per-round barrier and point swap cost k_iter_spec (timing builds through hipRTC flags; results wrong)
for F in "" "-DFL_X_HALF_BARRIERS" "-DFL_X_NOSWAP"; do for NS in 1024 1536; do
  FLAME_RTC_FLAGS="$F" FLAME_NSLOTS=$NS python3 bench.py --steps 10 --warmup 2 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('flags [%s] slots $NS: k_iter %.3f ms per frame (one lane), frame loop %.3f ms' % ('$F', k['iter'], d['ms_per_step']))"
done; done
