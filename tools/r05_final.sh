#!/bin/bash
# Round 5: the measurement artefacts kept under profiles/ (run on the GPU box; everything lands in gpurun_out/r05f_*).
#   cfg2: bench line, rocprofv3 kernel stats of the same command, TCC traffic, SQ counters (k_iter_spec, k_accum_tiles, k_de_dir 1/4/5/6)
#   cfg3 / cfg4 / cfg5: kernel stats, TCC traffic, SQ counters of iterate + accumulate, bench line; the DE slot budget (cfg2)
export TMPDIR=/tmp
O=gpurun_out
python3 bench.py > $O/r05f_bench.json 2> $O/r05f_bench.err
tools/prof_kernels.sh r05f_cfg2 > $O/r05f_cfg2_kernels.txt 2>&1
cp $O/prof_r05f_cfg2/bench_kernel_stats.csv $O/r05f_bench_kernel_stats.csv
BENCH_ARGS="" tools/pmc_traffic.sh r05f_cfg2 > $O/r05f_cfg2_traffic.txt 2>&1
tools/pmc_sq.sh r05f_de "k_de_dir<1" > $O/r05f_sq_de1.txt 2>&1
python3 - <<PY
import csv, collections, glob, json
for kern, out in (("k_de_dir<1", "k_de_dir1"), ("k_de_dir<4", "k_de_dir4"), ("k_de_dir<5", "k_de_dir5"), ("k_de_dir<6", "k_de_dir6"), ("k_iter_spec", "k_iter_spec"), ("k_accum_tiles", "k_accum_tiles")):
    per = collections.defaultdict(dict)
    for f in sorted(glob.glob("gpurun_out/sq_r05f_de_[0-9]/b_counter_collection.csv")):
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (c, d), v in acc.items():
            per[c][d] = v
    json.dump({c: sorted(d.values())[len(d) // 2] for c, d in per.items()}, open("gpurun_out/r05f_sq_counters_%s.json" % out, "w"), indent=1, sort_keys=True)
PY
tools/de_slot_budget.sh > /dev/null 2>&1
for cfg in cfg3 cfg4 cfg5; do
  export BENCH_ARGS="--config $cfg"
  tools/prof_kernels.sh r05f_$cfg --config $cfg > $O/r05f_${cfg}_kernels.txt 2>&1
  cp $O/prof_r05f_$cfg/bench_kernel_stats.csv $O/r05f_${cfg}_kernel_stats.csv
  tools/pmc_traffic.sh r05f_$cfg > $O/r05f_${cfg}_traffic.txt 2>&1
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
done
unset BENCH_ARGS
ls $O | grep r05f | head -50
# keep what goes to profiles/ (the summaries and the raw *_kernel_stats.csv copied above), drop the bulky per-pass trees: gpurun merges at most 64 MiB back
mkdir -p $O/keep
cp $O/r05f_* $O/keep/ 2>/dev/null
cp $O/pmc_r05f_*_traffic.json $O/sq_r05f_*.json $O/r05_de_slot_budget.txt $O/keep/ 2>/dev/null
find $O -mindepth 1 -maxdepth 1 ! -name keep -exec rm -rf {} +
mv $O/keep/* $O/ && rmdir $O/keep
ls $O
