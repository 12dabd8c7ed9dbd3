#!/bin/bash
# multi-launch frames: drains of launch k on the aux stream while launch k+1 iterates (default) vs strictly serial
for cfg in cfg3 cfg5; do for off in 0 1; do
  FLAME_NO_INTRA_OVERLAP=$off python bench.py --config $cfg --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1.5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('$cfg serial=$off: %.3f ms/frame  [one lane: iter %.3f accum+flush %.3f filt %.3f]' % (d['ms_per_step'], k['iter'], k['accum_flush'], k['filters']))"
done; done
