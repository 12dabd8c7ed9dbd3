#!/bin/bash
# usage: tools/prof_kernels.sh <tag> [bench args...]  -> prints per-kernel average durations
export TMPDIR=/tmp
tag=$1; shift
FLAME_LANES=1 FLAME_NO_INTRA_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --min-timed-frames 100 "$@" > gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_$tag/bench_kernel_stats.csv")))
for r in rows[:18]: print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), "%10.1f us avg"%(float(r["AverageNs"])/1e3), r["Percentage"])
PY
