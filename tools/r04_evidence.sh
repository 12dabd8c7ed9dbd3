#!/bin/bash
# Round 4: rocprofv3 evidence for one BASELINE config: kernel stats, TCC traffic, SQ counters of the iterate and
# accumulate kernels, and the bench line.  Usage: tools/r04_evidence.sh cfg5 [sq]   -> gpurun_out/r04_<cfg>_*
cfg=$1; sq=$2
export BENCH_ARGS="--config $cfg"
tools/prof_kernels.sh r04_$cfg --config $cfg > gpurun_out/r04_${cfg}_kernels.txt 2>&1
cp gpurun_out/prof_r04_$cfg/bench_kernel_stats.csv gpurun_out/r04_${cfg}_kernel_stats.csv
tools/pmc_traffic.sh r04_$cfg > gpurun_out/r04_${cfg}_traffic.txt 2>&1
if [ -n "$sq" ]; then
  tools/pmc_sq.sh r04_${cfg}_iter k_iter > gpurun_out/r04_${cfg}_sq_k_iter.txt 2>&1
  # the same passes hold every kernel's counters: second summary for the accumulate without re-running
  python3 - <<PY
import csv, collections, glob, json
acc = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/sq_r04_${cfg}_iter_*/b_counter_collection.csv")):
    tmp = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_accum_tiles" in r["Kernel_Name"]:
            tmp[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (c, d), v in tmp.items():
        acc[c][d] = v
out = {c: sorted(d.values())[len(d) // 2] for c, d in acc.items()}
json.dump(out, open("gpurun_out/sq_r04_${cfg}_accum.json", "w"), indent=1, sort_keys=True)
PY
fi
python3 bench.py --config $cfg --steps 6 --warmup 2 --cpu-seconds 0 --preheat-seconds 2 --min-timed-frames 24 > gpurun_out/r04_${cfg}_bench.json 2> gpurun_out/r04_${cfg}_bench.err
tail -c 600 gpurun_out/r04_${cfg}_bench.json
