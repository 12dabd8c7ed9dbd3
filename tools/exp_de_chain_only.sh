#!/bin/bash
# persistent DE launch with only ONE direction's tiles in the work lists (no waits build): its overhead per direction
# against the same tiles as a plain launch (FLAME_DE_CHAIN=2), rocprofv3 kernel times
export TMPDIR=/tmp
for only in 1 5 0; do
  FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip_nw.so FLAME_DE_CHAIN=1 FLAME_DE_CHAIN_ONLY=$only FLAME_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chonly_$only -o b -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 30 > gpurun_out/chonly_$only.log 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/chonly_$only/b_kernel_stats.csv")):
    if 'k_de_chain' in r["Name"]: print("only direction $only in the lists: k_de_chain %.1f us avg" % (float(r["AverageNs"])/1e3))
PY
done
FLAME_DE_CHAIN=2 FLAME_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chonly_ref -o b -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 30 > gpurun_out/chonly_ref.log 2>&1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/chonly_ref/b_kernel_stats.csv")) if 'k_de_one' in r["Name"]]
rows.sort(key=lambda r: r["Name"])
print("one direction per launch, same tiles:", " ".join("%s:%.1f" % (r["Name"].split("<")[1].split(">")[0], float(r["AverageNs"])/1e3) for r in rows), " sum %.1f us" % sum(float(r["AverageNs"])/1e3 for r in rows))
PY
FLAME_DE_CHAIN=1 FLAME_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chonly_all -o b -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 30 > gpurun_out/chonly_all.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/chonly_all/b_kernel_stats.csv")):
    if 'k_de_chain' in r["Name"] or 'k_ch_params' in r["Name"] or 'fill' in r["Name"].lower(): print(r["Name"][:50], "%.1f us avg" % (float(r["AverageNs"])/1e3), r["Calls"])
PY
