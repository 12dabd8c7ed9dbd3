#!/bin/bash
# Round 6: walker slots x rounds per batch on the frame loop (LDS per workgroup = 6 KB swap + 1 KB x rounds + 4.3 KB)
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
run() { python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1'.ljust(22), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'], d['roofline']['frac'])"; }
for rep in 1 2; do
  run default
  FLAME_NSLOTS=1280 FLAME_BIN_ROUNDS=12 run s1280r12
  FLAME_NSLOTS=1280 FLAME_BIN_ROUNDS=14 run s1280r14
  FLAME_NSLOTS=1536 FLAME_BIN_ROUNDS=10 run s1536r10
  FLAME_NSLOTS=1536 FLAME_BIN_ROUNDS=12 run s1536r12
  FLAME_NSLOTS=1024 FLAME_BIN_ROUNDS=12 run s1024r12
  FLAME_NSLOTS=1024 FLAME_BIN_ROUNDS=14 run s1024r14
done 2>&1 | tee gpurun_out/r06_twelfth.txt
