#!/bin/bash
# cfg3 (2^30 samples per frame, four launches): walker slots 1536 (six iterate workgroups per CU: their LDS leaves the other lane no room) against 1024 / 1280.
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --config cfg3 --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for rep in 1 2; do
  for n in ${SLOTS:-1536 1024 1280}; do
    export FLAME_NSLOTS=$n
    echo -n "== slots $n (rep $rep)  "
    python bench.py --config cfg3 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_frame'])"
  done
done
