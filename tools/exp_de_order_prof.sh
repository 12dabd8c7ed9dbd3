#!/bin/bash
export TMPDIR=/tmp
for m in "0 0" "0 1" "2 0"; do set -- $m
  FLAME_DE_CHAIN=$1 FLAME_DE_ORDER=$2 FLAME_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dep_$1$2 -o b -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 60 > gpurun_out/dep_$1$2.log 2>&1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/dep_$1$2/b_kernel_stats.csv")) if 'k_de_' in r["Name"]]
rows.sort(key=lambda r: r["Name"])
print("chain=$1 order=$2:", " ".join("%s:%.1f" % (r["Name"].split("<")[1].split(">")[0].replace(' ',''), float(r["AverageNs"])/1e3) for r in rows), " sum %.1f us" % sum(float(r["AverageNs"])/1e3 for r in rows))
PY
done
