#!/bin/bash
# 16-wave walker workgroups (batches of 16384 records) against the 8-wave default at 4K / 8K
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for cfg in cfg5 cfg4; do
for env in "FLAME_NW=8 FLAME_NSLOTS=1024" "FLAME_NW=16 FLAME_NSLOTS=1024" "FLAME_NW=16 FLAME_NSLOTS=1024 FLAME_BIN_ROUNDS=12" "FLAME_NW=16 FLAME_NSLOTS=1024 FLAME_BIN_ROUNDS=8"; do
  env $env python bench.py --config $cfg --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('$cfg $env: %.3f ms/frame  [iter %.3f accum+flush %.3f filt %.3f]' % (d['ms_per_step'], k['iter'], k['accum_flush'], k['filters']))"
done; done
