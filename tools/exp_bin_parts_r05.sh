#!/bin/bash
# round 5: workgroups per tile of k_accum_tiles (FLAME_BIN_PARTS) with the ganged tile order
# usage: tools/exp_bin_parts_r05.sh <config> <parts> [...]
export TMPDIR=/tmp FLAME_LANES=1 FLAME_NO_INTRA_OVERLAP=1
cfg=$1; shift
for p in "$@"; do
  FLAME_BIN_PARTS=$p rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bp_${cfg}_${p} -o b -- python3 bench.py --config $cfg --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 6 > gpurun_out/bp_${cfg}_${p}.log 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/bp_${cfg}_${p}/b_kernel_stats.csv")):
    if 'k_accum' in r["Name"]: print("$cfg parts=$p  k_accum_tiles %9.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
