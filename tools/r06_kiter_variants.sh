#!/bin/bash
# k_iter_spec alone under timing-experiment flags (some give wrong results: kernel time only, no bench line).
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for rep in 1 2; do
  i=0
  while IFS= read -r flags; do
    export FLAME_RTC_FLAGS="$flags"
    echo "== [$flags] (rep $rep)"
    tools/prof_kernels.sh kv_${i}_$rep --preheat-seconds 1.0 $BENCH_ARGS 2>&1 | grep -E "k_iter|k_accum" | head -2
    i=$((i+1))
  done < ${VARIANTS:-tools/r06_merge2.variants}
done
