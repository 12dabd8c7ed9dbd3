#!/bin/bash
# FL_ITER_MERGE A/B: the bit-exact tests under the flag, then k_iter_spec alone and the frame loop, alternating on one box.
export TMPDIR=/tmp
mkdir -p gpurun_out
M="-DFL_ITER_MERGE=1"
echo "== parity under $M"
FLAME_RTC_FLAGS="$M" timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variations.py tests/test_gpu_random_genomes.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for rep in 1 2; do
  for v in base merge; do
    if [ $v = merge ]; then export FLAME_RTC_FLAGS="$M"; else unset FLAME_RTC_FLAGS; fi
    echo "== $v (rep $rep)"
    tools/prof_kernels.sh merge_${v}_$rep --preheat-seconds 1.0 2>&1 | grep -E "k_iter|k_accum" | head -3
    python bench.py --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'])"
  done
done
for cfg in cfg3 cfg4; do
  for v in base merge; do
    if [ $v = merge ]; then export FLAME_RTC_FLAGS="$M"; else unset FLAME_RTC_FLAGS; fi
    echo "== $cfg $v"
    python bench.py --config $cfg --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'])"
  done
done
