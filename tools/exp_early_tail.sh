#!/bin/bash
# round 5 (late): the ten words behind a per-round-fetched record's head requested a round ahead with it (-DFL_EARLY_TAIL=1), against on demand
export TMPDIR=/tmp
run() { cfg=$1; n=$2; shift 2; env "$@" python bench.py --config $cfg --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms_per_frame']; r=d['roofline']
print('$cfg $*: %.3f ms/frame  chain %.4f  [iter %.3f accum+flush %.3f filt %.3f]' % (d['ms_per_step'], r['frac'], k['iter'], k['accum_flush'], k['filters']))"; }
FLAME_RTC_FLAGS="-DFL_EARLY_TAIL=1" timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -m gpu -k "per_genome or interpreter or paired_halves or cfg5_full or cfg4_full" 2>&1 | tail -2
for r in 1 2; do
  run cfg5 12 A=1; run cfg5 12 FLAME_RTC_FLAGS=-DFL_EARLY_TAIL=1
  run cfg4 60 A=1; run cfg4 60 FLAME_RTC_FLAGS=-DFL_EARLY_TAIL=1
  run cfg3 40 A=1; run cfg3 40 FLAME_RTC_FLAGS=-DFL_EARLY_TAIL=1
done
