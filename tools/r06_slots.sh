#!/bin/bash
# Walker slots below 1024 at cfg2 (three iterate workgroups per CU leave room for a workgroup of the other lane's accumulate): frame loop, one box.
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for rep in 1 2; do
  for n in 1024 768 896 512; do
    export FLAME_NSLOTS=$n
    echo "== slots $n (rep $rep)"
    python bench.py --cpu-seconds 0 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_frame'])"
  done
done
