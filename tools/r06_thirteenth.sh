#!/bin/bash
# Round 6: the paired 8-wave walker geometry (512 workgroups, two temporal samples each) at 1080p, with the lean accumulate
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
run() { python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1'.ljust(22), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'], d['roofline']['frac'], d['config']['walker_waves'], d['config']['walker_slots'])"; }
for rep in 1 2; do
  run default
  FLAME_NW=8 FLAME_NSLOTS=512 run w8s512r16
  FLAME_NW=8 FLAME_NSLOTS=512 FLAME_BIN_ROUNDS=12 run w8s512r12
  FLAME_NW=8 FLAME_NSLOTS=512 FLAME_BIN_ROUNDS=8 run w8s512r8
  FLAME_NW=8 FLAME_NSLOTS=512 FLAME_BIN_PARTS=8 run w8s512r16p8
done 2>&1 | tee gpurun_out/r06_thirteenth.txt
