#!/bin/bash
# workgroups per tile after the palette moved to LDS (chunked by palette rows)
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
run() { env FLAME_BIN_PARTS=$2 python bench.py --config $1 --steps 6 --warmup 1 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('$1 parts=$2: %.3f ms/frame  [iter %.3f accum+flush %.3f filt %.3f]' % (d['ms_per_step'], k['iter'], k['accum_flush'], k['filters']))"; }
for P in 12 16 20; do run cfg2 $P; done
for P in 5 7 10 13 16; do run cfg4 $P; done
for P in 6 8 10 12; do run cfg5 $P; done
