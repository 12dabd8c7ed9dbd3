#!/bin/bash
# Round 6: geometry switches on the frame loop with the lean accumulate (walker slots, stream lanes, queue depth)
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
run() { python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1'.ljust(22), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'], d['roofline']['frac'])"; }
for rep in 1 2; do
  run default ""
  FLAME_NSLOTS=1536 run slots1536 ""
  FLAME_NSLOTS=1280 run slots1280 ""
  FLAME_LANES=3 run lanes3 ""
  run depth3 "--depth 3"
  FLAME_LANES=3 run lanes3depth3 "--depth 3"
done 2>&1 | tee gpurun_out/r06_eleventh.txt
