#!/bin/bash
# Round 6: cfg2 with 16-wave workgroups in quarters (256 slots), and cfg3 geometry
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
run() { python bench.py --config $2 --steps $3 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 --min-timed-frames $4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2 $1'.ljust(22), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'], d['roofline']['frac'], d['config']['walker_waves'], d['config']['walker_slots'])"; }
for rep in 1 2; do
  run default cfg2 8 300
  FLAME_NW=16 FLAME_NSLOTS=256 run w16s256 cfg2 8 300
  FLAME_NW=16 FLAME_NSLOTS=256 FLAME_BIN_PARTS=8 run w16s256p8 cfg2 8 300
  run default cfg3 6 100
  FLAME_NW=8 FLAME_NSLOTS=1024 run w8s1024 cfg3 6 100
  FLAME_NW=16 FLAME_NSLOTS=1024 run w16s1024 cfg3 6 100
done 2>&1 | tee gpurun_out/r06_cfg2_w16.txt
