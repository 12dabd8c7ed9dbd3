#!/bin/bash
# Round 5: cfg2's bench line, kernel stats and TCC traffic again (the headline artefacts of tools/r05_final.sh without the counter passes and the slot budget)
export TMPDIR=/tmp
O=gpurun_out
tools/prof_kernels.sh r05f_cfg2 > $O/r05f_cfg2_kernels.txt 2>&1
cp $O/prof_r05f_cfg2/bench_kernel_stats.csv $O/r05f_bench_kernel_stats.csv
BENCH_ARGS="" tools/pmc_traffic.sh r05f_cfg2 > $O/r05f_cfg2_traffic.txt 2>&1
cp $O/pmc_r05f_cfg2_traffic.json profiles/r05_pmc_traffic.json
python3 bench.py > $O/r05f_bench.json 2> $O/r05f_bench.err
mkdir -p $O/keep
cp $O/r05f_* $O/pmc_r05f_*_traffic.json $O/keep/ 2>/dev/null
find $O -mindepth 1 -maxdepth 1 ! -name keep -exec rm -rf {} +
mv $O/keep/* $O/ && rmdir $O/keep
ls $O; tail -c 400 $O/r05f_bench.json
