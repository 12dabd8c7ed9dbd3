#!/bin/bash
# DE kernels under several builds of the library (tile shapes): parity tests + rocprofv3 averages, one line per build
# usage: tools/ab_de.sh <lib> [<lib> ...]
export TMPDIR=/tmp
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 --min-timed-frames 0 > /dev/null 2>&1
for L in "$@"; do
  export FLAME_HIP_LIB=$PWD/$L
  t=$(python -m pytest tests/test_gpu_parity.py -q -k "bilateral or de_fused or deferred" 2>&1 | tail -1)
  tag=$(basename $L .so)
  FLAME_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abde_$tag -o b -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 60 > gpurun_out/abde_$tag.log 2>&1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/abde_$tag/b_kernel_stats.csv")) if 'k_de_' in r["Name"]]
rows.sort(key=lambda r: r["Name"])
print("$tag".ljust(20), " ".join("%s:%.1f" % (r["Name"].split("<")[1].split(",")[0], float(r["AverageNs"])/1e3) for r in rows), " sum %.1f us" % sum(float(r["AverageNs"])/1e3 for r in rows), " | $t")
PY
done
