#!/bin/bash
# regenerate the bench lines kept under profiles/ with the final build (the traffic files must be in profiles/ already)
export TMPDIR=/tmp
python3 bench.py > gpurun_out/r04g_bench.json 2> gpurun_out/r04g_bench.err
tools/prof_kernels.sh r04g_cfg2 > gpurun_out/r04g_cfg2_kernels.txt 2>&1
cp gpurun_out/prof_r04g_cfg2/bench_kernel_stats.csv gpurun_out/r04g_bench_kernel_stats.csv
for cfg in cfg3 cfg4 cfg5; do
  python3 bench.py --config $cfg --steps 6 --warmup 2 --cpu-seconds 0 --preheat-seconds 2 --min-timed-frames 24 > gpurun_out/r04g_${cfg}_bench.json 2> gpurun_out/r04g_${cfg}_bench.err
done
