#!/bin/bash
# round 5 (late): binned launches capped at 1024 rounds (rounds 1-5) against the reference's own schedule 1024, 1536, 2304 (cap 2304)
export TMPDIR=/tmp
run() { cfg=$1; shift; env "$@" python bench.py --config $cfg --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 24 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms_per_frame']; r=d['roofline']
print('$cfg $*: %.3f ms/frame  chain %.4f  [iter %.3f accum+flush %.3f filt %.3f]' % (d['ms_per_step'], r['frac'], k['iter'], k['accum_flush'], k['filters']))"; }
for c in cfg5 cfg3; do for r in 1 2; do run $c FLAME_LAUNCH_ROUNDS=1024; run $c FLAME_LAUNCH_ROUNDS=2304; done; done
