#!/bin/bash
# 8K accumulate: records in flight per lane (library variants) and workgroups per tile
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
run() { env $1 FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/$2 python bench.py --config ${3:-cfg5} --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('${3:-cfg5} $1 $2: %.3f ms/frame  [iter %.3f accum+flush %.3f filt %.3f]' % (d['ms_per_step'], k['iter'], k['accum_flush'], k['filters']))"; }
for P in 8 12 16; do for L in libflame_hip.so libflame_hip_w8.so; do run "FLAME_BIN_PARTS=$P" $L; done; done
for P in 4 7 12 16; do run "FLAME_BIN_PARTS=$P" libflame_hip.so cfg4; done
for P in 8 16 24 30; do run "FLAME_BIN_PARTS=$P" libflame_hip.so cfg2; done
