#!/bin/bash
# Round 5: the per-config artefacts of tools/r05_final.sh for ONE config again (after a change that only that size class sees), bench line included.
# usage: tools/r05_final_cfg.sh cfg5        (then tools/r05_collect.sh copies them into profiles/)
export TMPDIR=/tmp
O=gpurun_out
cfg=$1
export BENCH_ARGS="--config $cfg"
tools/prof_kernels.sh r05f_$cfg --config $cfg > $O/r05f_${cfg}_kernels.txt 2>&1
cp $O/prof_r05f_$cfg/bench_kernel_stats.csv $O/r05f_${cfg}_kernel_stats.csv
tools/pmc_traffic.sh r05f_$cfg > $O/r05f_${cfg}_traffic.txt 2>&1
cp $O/pmc_r05f_${cfg}_traffic.json profiles/r05_${cfg}_pmc_traffic.json          # (the bench line reads the newest committed traffic file)
tools/pmc_sq.sh r05f_${cfg}_iter k_iter > $O/r05f_${cfg}_sq_k_iter.txt 2>&1
python3 - <<PY
import csv, collections, glob, json
acc = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/sq_r05f_${cfg}_iter_*/b_counter_collection.csv")):
    tmp = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_accum_tiles" in r["Kernel_Name"]:
            tmp[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (c, d), v in tmp.items():
        acc[c][d] = v
json.dump({c: sorted(d.values())[len(d) // 2] for c, d in acc.items()}, open("gpurun_out/sq_r05f_${cfg}_accum.json", "w"), indent=1, sort_keys=True)
PY
cp $O/sq_r05f_${cfg}_iter.json profiles/r05_${cfg}_sq_counters_k_iter_spec.json
unset BENCH_ARGS
python3 bench.py --config $cfg > $O/r05f_${cfg}_bench.json 2> $O/r05f_${cfg}_bench.err
mkdir -p $O/keep
cp $O/r05f_* $O/keep/ 2>/dev/null
cp $O/pmc_r05f_*_traffic.json $O/sq_r05f_*.json $O/keep/ 2>/dev/null
find $O -mindepth 1 -maxdepth 1 ! -name keep -exec rm -rf {} +
mv $O/keep/* $O/ && rmdir $O/keep
ls $O; tail -c 600 $O/r05f_${cfg}_bench.json
