#!/bin/bash
# The small-image geometry of frames of more than 2^28 samples: 1536 walker slots (round 3's choice) against 1280 and 1024, by flame and sample count.
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --config cfg3 --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for case in "cfg3 536870912" "cfg3 1073741824" "cfg3 2147483648" "cfg2 536870912" "cfg2 1073741824" "cfg2 4294967296"; do
  set -- $case
  for n in 1536 1280 1024; do
    export FLAME_NSLOTS=$n FLAME_BENCH_SAMPLES=$2
    echo -n "== $1 samples $2 slots $n  "
    python bench.py --config $1 --cpu-seconds 0 --min-timed-frames 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
