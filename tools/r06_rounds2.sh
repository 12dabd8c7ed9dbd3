#!/bin/bash
# Rounds per sorted batch below 16 (less LDS per iterate workgroup: room for the other lane's accumulate beside it?): frame loop, one box.
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for rep in 1 2; do
  for r in 16 12 10 8 6; do
    export FLAME_BIN_ROUNDS=$r
    echo "== rounds $r (rep $rep)"
    python bench.py --cpu-seconds 0 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
