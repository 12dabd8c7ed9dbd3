#!/bin/bash
for p in 0 6 10 12 16; do
  if [ $p = 0 ]; then unset FLAME_BIN_PARTS; else export FLAME_BIN_PARTS=$p; fi
  python bench.py --config cfg5 --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('cfg5 parts $p: %.3f ms/frame [iter %.3f accum+flush %.3f filters %.3f]' % (d['ms_per_step'], k['iter'], k['accum_flush'], k['filters']))"
done
for p in 0 8 12 24 32; do
  if [ $p = 0 ]; then unset FLAME_BIN_PARTS; else export FLAME_BIN_PARTS=$p; fi
  python bench.py --config cfg3 --steps 6 --warmup 2 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 24 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('cfg3 parts $p: %.3f ms/frame [iter %.3f accum+flush %.3f filters %.3f]' % (d['ms_per_step'], k['iter'], k['accum_flush'], k['filters']))"
done
