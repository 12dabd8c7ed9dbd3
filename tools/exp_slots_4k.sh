#!/bin/bash
# round 5 (late): fewer walker slots at 4K / 8K — every walker spends the reference's 256 un-plotted rounds per frame, and only
# 2 (4K, 57 KB) / 1 (8K, 137 KB) workgroups are resident per CU anyway: 512 slots of 8 waves are all resident at once at 4K
# usage: tools/exp_slots_4k.sh <config> <nw> <nslots> [<nslots> ...]
export TMPDIR=/tmp
cfg=$1; nw=$2; shift 2
run() { env "$@" python bench.py --config $cfg --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms_per_frame']; r=d['roofline']
print('$cfg $*: %.3f ms/frame  chain %.3f  [iter %.3f accum+flush %.3f filt %.3f]' % (d['ms_per_step'], r['frac'], k['iter'], k['accum_flush'], k['filters']))"; }
for s in "$@"; do run FLAME_NW=$nw FLAME_NSLOTS=$s; done
