#!/bin/bash
# round 5 (late): 512 slots of 8 waves with paired halves (two temporal samples per workgroup) against 1024 slots of 8 waves, cfg4
export TMPDIR=/tmp
run() { python bench.py --config cfg4 --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms_per_frame']; r=d['roofline']
print('cfg4 $1: %.3f ms/frame  chain %.4f  [iter %.3f accum+flush %.3f filt %.3f]  slots %s x %s waves' % (d['ms_per_step'], r['frac'], k['iter'], k['accum_flush'], k['filters'], d['config'].get('walker_slots'), d['config'].get('walker_waves')))"; }
for r in 1 2 3; do
  FLAME_NW=8 FLAME_NSLOTS=1024 run "8x1024"
  FLAME_NW=8 FLAME_NSLOTS=512 run "8x512 paired"
  run "auto"
done
