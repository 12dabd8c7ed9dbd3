#!/bin/bash
# walker geometry x scan variant (libflame_hip_x.so: tile-count scan always by all waves)
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for cfg in cfg2 cfg4; do
for env in "FLAME_NW=4 FLAME_NSLOTS=1536" "FLAME_NW=8 FLAME_NSLOTS=1024" "FLAME_NW=16 FLAME_NSLOTS=1024"; do
for L in libflame_hip.so libflame_hip_x.so; do
  env $env FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/$L python bench.py --config $cfg --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('$cfg $env $L: %.3f ms/frame  [iter %.3f accum+flush %.3f filt %.3f]' % (d['ms_per_step'], k['iter'], k['accum_flush'], k['filters']))"
done; done; done
