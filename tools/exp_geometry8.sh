#!/bin/bash
# 8-wave workgroups at 1080p with short batches: 4 workgroups per CU = 32 waves per CU (8 per SIMD)
run() { env "$@" python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --preheat-seconds 1.5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('$*: %.3f ms/frame  [iter %.3f accum+flush %.3f filt %.3f]' % (d['ms_per_step'], k['iter'], k['accum_flush'], k['filters']))"; }
run FLAME_NSLOTS=1536 FLAME_BIN_ROUNDS=16
run FLAME_NW=8 FLAME_NSLOTS=1024 FLAME_BIN_ROUNDS=8
run FLAME_NW=8 FLAME_NSLOTS=1024 FLAME_BIN_ROUNDS=10
run FLAME_NW=8 FLAME_NSLOTS=1024 FLAME_BIN_ROUNDS=12
run FLAME_NW=8 FLAME_NSLOTS=1024 FLAME_BIN_ROUNDS=16
run FLAME_NW=8 FLAME_NSLOTS=1280 FLAME_BIN_ROUNDS=8
