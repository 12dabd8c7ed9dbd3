#!/bin/bash
# cfg5 (8K, 16-wave workgroups of 137 KB LDS: nothing runs beside them): fewer rounds per sorted batch leave LDS for the other lane's DE — does the frame gain?
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --config cfg5 --steps 2 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 --min-timed-frames 0 > /dev/null 2>&1
for r in ${ROUNDS:-16 14 12 10}; do
  export FLAME_BIN_ROUNDS=$r
  echo -n "== rounds $r  "
  python bench.py --config cfg5 --cpu-seconds 0 --steps 12 --warmup 2 --min-timed-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_frame'])"
done
