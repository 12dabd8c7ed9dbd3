#!/bin/bash
# round 5: DE tile order per direction and image size — 0 per-XCD column-major runs, 1 plain row-major, 2 row-major in runs of DE_RUN per XCD
# usage: tools/exp_de_order2.sh <config> <min-timed-frames> <order> [<order> ...]       (order: one digit or eight, FLAME_DE_ORDER)
export TMPDIR=/tmp
cfg=$1; n=$2; shift 2
for o in "$@"; do
  FLAME_DE_ORDER=$o FLAME_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/deo_${cfg}_$o -o b -- python3 bench.py --config $cfg --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames $n > gpurun_out/deo_${cfg}_$o.log 2>&1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/deo_${cfg}_$o/b_kernel_stats.csv")) if 'k_de_' in r["Name"]]
rows.sort(key=lambda r: r["Name"])
print("$cfg order=$o:", " ".join("%s:%.1f" % (r["Name"].split("<")[1].split(",")[0], float(r["AverageNs"])/1e3) for r in rows), " sum %.1f us" % sum(float(r["AverageNs"])/1e3 for r in rows))
PY
done
