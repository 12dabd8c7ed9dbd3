#!/bin/bash
# (round 5: FLAME_FLUSH_LAST, which this experiment ran with in round 4, no longer exists: one flush per launch)
# Round 4: keep the sample log on die?  A frame's iterate / accumulate pair cut into launches of FLAME_LAUNCH_ROUNDS
# write-enabled rounds (log chunk = nslots x 256 x rounds x 4 B: 1024 rounds = 1 GiB, 96 = 96 MB < the 256 MiB Infinity
# Cache), launch k+1 iterating while launch k is accumulated (two log sets), one flush at the end, log stores
# non-temporal or not, workgroups per tile of the accumulate.  Frame-loop ms and one-lane kernel sums per setting.
# usage: tools/exp_log_chunks.sh ["ROUNDS NT PARTS" ...]
for S in "$@"; do
  set -- $S; R=$1; NT=$2; P=$3
  env FLAME_LAUNCH_ROUNDS=$R FLAME_RTC_FLAGS=-DFL_LOG_NT=$NT FLAME_BIN_PARTS=$P \
    python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; r=d['roofline']
print('rounds/launch %5s  log-nt %s  parts %2s: frame %.3f ms  [one lane: iter %.3f  accum+flush %.3f (accum %.3f)  chain %.3f ms = %.3f of 8 TB/s]' % ('$R', '$NT', '$P', d['ms_per_step'], k['iter'], k['accum_flush'], r['k_accum_tiles_ms_per_frame'], r['chain_ms_per_frame'], r['frac']))"
done
