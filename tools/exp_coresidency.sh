#!/bin/bash
# Round 4: an accumulate that fits beside the walkers — 128 x 32-pixel tiles (46 KB of LDS per workgroup instead of 80) —
# A/B on the FRAME LOOP (bench.py), default build against -DFL_TILE_H_LOG2=5 with 1024- and 512-thread accumulate workgroups
run() { # lib rtcflags label
  FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/$1 FLAME_RTC_FLAGS="$2" python3 bench.py --steps 30 --warmup 3 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 150 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; r=d['roofline']
print('$3: frame loop %.3f ms  [one lane: iter %.3f  accum %.3f  filters %.3f; sum %.3f]' % (d['ms_per_step'], k['iter'], r['k_accum_tiles_ms_per_frame'], k['filters'], d['config']['stream_lanes']['sum_of_big_kernels_ms']))"
}
for i in 1 2; do
  run libflame_hip.so "" "128x64 tiles, 2 x 1024-thread accumulate workgroups per CU (default)"
  run libflame_hip_t5.so "-DFL_TILE_H_LOG2=5" "128x32 tiles, 1024-thread accumulate workgroups"
  run libflame_hip_t5h.so "-DFL_TILE_H_LOG2=5" "128x32 tiles, 512-thread accumulate workgroups"
done
FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip_t5.so FLAME_RTC_FLAGS="-DFL_TILE_H_LOG2=5" python -m pytest tests/test_gpu_parity.py -q -x -k "binned or iter_bit_exact or larger_workgroups" 2>&1 | tail -2
