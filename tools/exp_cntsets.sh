#!/bin/bash
# sets of tile counters in k_iter (library variants built with -DFL_CNT_SETS=n; _a = the previous build)
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
run() { env $2 FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/$1 python bench.py --config ${3:-cfg2} --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('${3:-cfg2} $1 $2: %.3f ms/frame  [iter %.3f accum+flush %.3f filt %.3f]' % (d['ms_per_step'], k['iter'], k['accum_flush'], k['filters']))"; }
for c in cfg2 cfg3 cfg4 cfg5; do for i in 1 2; do
run libflame_hip_a.so FLAME_X=0 $c
run libflame_hip_s2.so FLAME_X=0 $c
run libflame_hip.so FLAME_X=0 $c
done; done
