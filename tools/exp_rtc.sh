#!/bin/bash
# frame time of the pipelined bench loop under a few switches
run() { env "$@" python bench.py --steps 40 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('$*: %.1f Ms/s  %.3f ms/frame  [iter %.3f accum %.3f filt %.3f]' % (d['value'], d['ms_per_step'], k['iter'], k['accum_flush'], k['filters']))"; }
run FLAME_RTC=1 FLAME_LANES=2
run FLAME_RTC=0 FLAME_LANES=2
