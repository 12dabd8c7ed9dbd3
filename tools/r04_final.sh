#!/bin/bash
# Round 4: the measurement artefacts kept under profiles/ (run on the GPU box; everything lands in gpurun_out/r04f_*)
export TMPDIR=/tmp
O=gpurun_out
python3 bench.py > $O/r04f_bench.json 2> $O/r04f_bench.err
tools/prof_kernels.sh r04f_cfg2 > $O/r04f_cfg2_kernels.txt 2>&1
cp $O/prof_r04f_cfg2/bench_kernel_stats.csv $O/r04f_bench_kernel_stats.csv
BENCH_ARGS="" tools/pmc_traffic.sh r04f_cfg2 > $O/r04f_cfg2_traffic.txt 2>&1
tools/pmc_sq.sh r04f_de "k_de_dir<1" > $O/r04f_sq_de1.txt 2>&1
python3 - <<PY
import csv, collections, glob, json
for tag, kern, out in (("r04f_de", "k_de_dir<5", "r04f_sq_k_de_dir5.json"), ("r04f_de", "k_iter_spec", "r04f_sq_k_iter_spec.json"), ("r04f_de", "k_accum_tiles", "r04f_sq_k_accum_tiles.json")):
    per = collections.defaultdict(dict)
    for f in sorted(glob.glob("gpurun_out/sq_%s_[0-9]/b_counter_collection.csv" % tag)):
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (c, d), v in acc.items():
            per[c][d] = v
    json.dump({c: sorted(d.values())[len(d) // 2] for c, d in per.items()}, open("gpurun_out/" + out, "w"), indent=1, sort_keys=True)
PY
for cfg in cfg3 cfg4 cfg5; do
  tools/r04_evidence.sh $cfg sq > /dev/null 2>&1
done
