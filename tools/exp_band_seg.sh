#!/bin/bash
# band walker of the DE under several segment lengths (FLAME_DE_SEG_ROWS) and the tile form: rocprofv3 averages per direction
export TMPDIR=/tmp
for v in tiles auto "$@"; do
  unset FLAME_DE_BAND FLAME_DE_SEG_ROWS
  if [ "$v" != tiles ]; then export FLAME_DE_BAND=1; fi
  if [ "$v" != tiles ] && [ "$v" != auto ]; then export FLAME_DE_SEG_ROWS=$v; fi
  FLAME_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/band_$v -o b -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --min-timed-frames 60 > gpurun_out/band_$v.log 2>&1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/band_$v/b_kernel_stats.csv")) if 'k_de_' in r["Name"]]
rows.sort(key=lambda r: r["Name"])
print("$v".ljust(6), " ".join("%s:%.1f" % (r["Name"].split("<")[1].split(",")[0], float(r["AverageNs"])/1e3) for r in rows), " sum %.1f us" % sum(float(r["AverageNs"])/1e3 for r in rows))
PY
done
