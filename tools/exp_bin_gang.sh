#!/bin/bash
# round 5: k_accum_tiles with G adjacent tiles of one part ganged on one XCD (FLAME_BIN_GANG): time and L2 fetch per launch
# usage: tools/exp_bin_gang.sh <config> <gang> [<gang> ...]
export TMPDIR=/tmp FLAME_LANES=1 FLAME_NO_INTRA_OVERLAP=1
cfg=$1; shift
for g in "$@"; do
  export FLAME_BIN_GANG=$g
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bg_${cfg}_${g}_t -o b -- python3 bench.py --config $cfg --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 6 > gpurun_out/bg_${cfg}_${g}_t.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/bg_${cfg}_${g}_c -o b -- python3 bench.py --config $cfg --steps 2 --warmup 1 --cpu-seconds 0 --preheat-seconds 0 --min-timed-frames 0 > gpurun_out/bg_${cfg}_${g}_c.log 2>&1
  python3 - <<PY
import csv, collections
t = {r["Name"].split("(")[0][:24]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open("gpurun_out/bg_${cfg}_${g}_t/b_kernel_stats.csv"))}
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("gpurun_out/bg_${cfg}_${g}_c/b_counter_collection.csv")):
    per[r["Kernel_Name"].split("(")[0][:24]][r["Dispatch_Id"]] += float(r["Counter_Value"])
def med(k):
    v = sorted(per[k].values()); return v[len(v) // 2] if v else 0.0
for k in t:
    if k.startswith("k_accum") or k.startswith("k_iter"):
        print("$cfg gang=$g %-24s %9.1f us   FETCH_SIZE %10.0f KB x2 = %7.3f GB per launch" % (k, t[k], med(k), med(k) * 2 * 1024 / 1e9))
PY
done
