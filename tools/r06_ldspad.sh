#!/bin/bash
# Unused LDS added to every iterate workgroup (FLAME_X_ITER_LDS_PAD): where does the two-lane frame loop stop fitting?  Frame loop + kernels alone.
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for rep in 1 2; do
  for pad in ${PADS:-0 512 1024 1536 2048 3072 4096 6144 8192 12288}; do
    export FLAME_X_ITER_LDS_PAD=$pad
    echo -n "== pad $pad (rep $rep)  "
    python bench.py --cpu-seconds 0 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_frame']['iter'])"
  done
done
