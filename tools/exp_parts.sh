#!/bin/bash
# workgroups per tile beyond 16 (explicit FLAME_BIN_PARTS lifts the default cap): is the accumulate tail-bound?
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for cfg in cfg2 cfg3; do for P in 16 24 32 48 64; do
  FLAME_BIN_PARTS=$P python bench.py --config $cfg --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('$cfg parts=$P: %.3f ms/frame  [iter %.3f accum+flush %.3f filt %.3f]' % (d['ms_per_step'], k['iter'], k['accum_flush'], k['filters']))"
done; done
