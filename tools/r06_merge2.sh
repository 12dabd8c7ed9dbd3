#!/bin/bash
# A/B of k_iter_spec code-generation variants through FLAME_RTC_FLAGS, alternating on one box: kernel alone + frame loop.
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for rep in 1 2 3; do
  i=0
  while IFS= read -r flags; do
    export FLAME_RTC_FLAGS="$flags"
    echo "== [$flags] (rep $rep)"
    tools/prof_kernels.sh m2_${i}_$rep --preheat-seconds 1.0 $BENCH_ARGS 2>&1 | grep -E "k_iter" | head -1
    python bench.py --cpu-seconds 0 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'])"
    i=$((i+1))
  done < ${VARIANTS:-tools/r06_merge2.variants}
done
