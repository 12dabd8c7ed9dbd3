#!/bin/bash
# Round 6: cfg4 with 16-wave workgroups in quarters (256 slots) x workgroups per tile
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
run() { python bench.py --config cfg4 --steps 6 --warmup 1 --cpu-seconds 0 --preheat-seconds 1 --min-timed-frames 120 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1'.ljust(22), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'], d['roofline']['frac'], d['config']['walker_waves'], d['config']['walker_slots'])"; }
for rep in 1 2; do
  run default
  for p in 3 4 6 8; do FLAME_NW=16 FLAME_NSLOTS=256 FLAME_BIN_PARTS=$p run w16s256p$p; done
  FLAME_BIN_PARTS=3 run w8s512p3
  FLAME_BIN_PARTS=5 run w8s512p5
done 2>&1 | tee gpurun_out/r06_cfg4_geom2.txt
