#!/bin/bash
# Round 4: DE workgroups per CU (FLAME_DE_WGS="r0,..,r7", 0 = the kernel's maximum; unset = the launcher's choice):
# rocprofv3 averages per direction for each setting, one line per setting.  usage: tools/exp_de_residency.sh "<setting>" ...
export TMPDIR=/tmp
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 --min-timed-frames 0 > /dev/null 2>&1
i=0
for S in "$@"; do
  if [ "$S" = auto ]; then unset FLAME_DE_WGS; else export FLAME_DE_WGS=$S; fi
  FLAME_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/deres_$i -o b -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 60 > gpurun_out/deres_$i.log 2>&1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/deres_$i/b_kernel_stats.csv")) if 'k_de_' in r["Name"]]
rows.sort(key=lambda r: r["Name"])
print("$S".ljust(18), " ".join("%s:%.1f" % (r["Name"].split("<")[1].split(",")[0], float(r["AverageNs"])/1e3) for r in rows), " sum %.1f us" % sum(float(r["AverageNs"])/1e3 for r in rows))
PY
  i=$((i+1))
done
