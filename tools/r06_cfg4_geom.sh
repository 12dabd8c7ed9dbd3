#!/bin/bash
# Round 6: walker geometry for cfg4 (4K, 2^28 samples per frame, 1086 tiles of 128x64) with the packed log
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
run() { python bench.py --config cfg4 --steps 6 --warmup 1 --cpu-seconds 0 --preheat-seconds 1 --min-timed-frames 120 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1'.ljust(22), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'], d['roofline']['frac'], d['config']['walker_waves'], d['config']['walker_slots'])"; }
for rep in 1 2; do
  run default
  FLAME_NW=16 FLAME_NSLOTS=256 run w16s256
  FLAME_NW=8 FLAME_NSLOTS=1024 run w8s1024
  FLAME_NW=4 FLAME_NSLOTS=1024 run w4s1024
  FLAME_BIN_PARTS=8 run parts8
  FLAME_BIN_PARTS=4 run parts4
done 2>&1 | tee gpurun_out/r06_cfg4_geom.txt
