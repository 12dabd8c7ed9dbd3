#!/bin/bash
# Round 6: the measurement artefacts kept under profiles/ (run on the GPU box through gpurun; everything that is kept lands in gpurun_out/r06f/
# under its profiles/ name).  Counter files first — they go into the box's profiles/ at once, so that the bench line that follows carries
# roofline.traffic and the counter-derived fractions of THIS library (bench.py quotes a counter file only when its sha256 matches).
#   cfg2: TCC traffic, SQ counters (k_iter_spec, k_accum_tiles_p3, k_de_dir 1/4/5/6), DE slot budget, rocprofv3 kernel stats, the bench line,
#         the bench line and kernel stats of the direct-atomic back-end (--accum atomic)
#   cfg3 / cfg4 / cfg5: TCC traffic, SQ counters of iterate + accumulate, kernel stats, bench line
# usage: tools/r06_final.sh [cfg2] [cfg3] [cfg4] [cfg5] [atomic]     (default: all)
export TMPDIR=/tmp
O=gpurun_out
K=$O/r06f
mkdir -p $K
what="$*"; [ -z "$what" ] && what="cfg2 cfg3 cfg4 cfg5 atomic"
sq_json() {   # sq_json <pass tag> <kernel substring> <output file>: medians per launch of every counter of the passes gpurun_out/sq_<tag>_N
python3 - "$1" "$2" "$3" <<'PY'
import csv, collections, glob, hashlib, json, os, sys
tag, kern, out = sys.argv[1:4]
per = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/sq_%s_[0-9]/b_counter_collection.csv" % tag)):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (c, d), v in acc.items():
        per[c][d] = v
res = {c: sorted(d.values())[len(d) // 2] for c, d in per.items()}
lib = os.environ.get("FLAME_HIP_LIB", "cuburn_amd/_lib/libflame_hip.so")
res["_lib_sha256"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
PY
}
python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 3 > /dev/null 2>&1
for cfg in $what; do
  case $cfg in
  cfg2)
    BENCH_ARGS="" tools/pmc_traffic.sh r06f_cfg2 > $K/r06_cfg2_traffic.txt 2>&1
    cp $O/pmc_r06f_cfg2_traffic.json profiles/r06_pmc_traffic.json; cp profiles/r06_pmc_traffic.json $K/
    tools/pmc_sq.sh r06f_cfg2 "k_iter_spec" > /dev/null 2>&1
    sq_json r06f_cfg2 k_iter_spec profiles/r06_sq_counters_k_iter_spec.json
    sq_json r06f_cfg2 k_accum_tiles profiles/r06_sq_counters_k_accum_tiles.json
    for d in 1 4 5 6; do sq_json r06f_cfg2 "k_de_dir<$d" profiles/r06_sq_counters_k_de_dir$d.json; done
    cp profiles/r06_sq_counters_*.json $K/
    DE_BUDGET_TAG=r06 tools/de_slot_budget.sh > /dev/null 2>&1
    cp $O/r06_de_slot_budget.txt profiles/r06_de_slot_budget.txt; cp profiles/r06_de_slot_budget.txt $K/
    tools/prof_kernels.sh r06f_cfg2 > $K/r06_cfg2_kernels.txt 2>&1
    cp $O/prof_r06f_cfg2/bench_kernel_stats.csv $K/r06_bench_kernel_stats.csv
    python3 bench.py > $K/r06_bench.json 2> $K/r06_bench.err
    ;;
  atomic)
    tools/prof_kernels.sh r06f_atomic --accum atomic --min-timed-frames 40 > $K/r06_atomic_kernels.txt 2>&1
    cp $O/prof_r06f_atomic/bench_kernel_stats.csv $K/r06_atomic_kernel_stats.csv
    python3 bench.py --accum atomic --steps 10 --warmup 2 --min-timed-frames 40 --cpu-seconds 0 > $K/r06_atomic_bench.json 2> $K/r06_atomic_bench.err
    ;;
  cfg3|cfg4|cfg5)
    export BENCH_ARGS="--config $cfg"
    tools/pmc_traffic.sh r06f_$cfg > $K/r06_${cfg}_traffic.txt 2>&1
    cp $O/pmc_r06f_${cfg}_traffic.json profiles/r06_${cfg}_pmc_traffic.json; cp profiles/r06_${cfg}_pmc_traffic.json $K/
    tools/pmc_sq.sh r06f_$cfg k_iter > /dev/null 2>&1
    sq_json r06f_$cfg k_iter profiles/r06_${cfg}_sq_counters_k_iter_spec.json
    sq_json r06f_$cfg k_accum_tiles profiles/r06_${cfg}_sq_counters_k_accum_tiles.json
    cp profiles/r06_${cfg}_sq_counters_*.json $K/
    unset BENCH_ARGS
    tools/prof_kernels.sh r06f_$cfg --config $cfg > $K/r06_${cfg}_kernels.txt 2>&1
    cp $O/prof_r06f_$cfg/bench_kernel_stats.csv $K/r06_${cfg}_kernel_stats.csv
    python3 bench.py --config $cfg --cpu-seconds 0 > $K/r06_${cfg}_bench.json 2> $K/r06_${cfg}_bench.err
    ;;
  esac
done
# keep the summaries, drop the bulky per-pass trees: gpurun merges at most 64 MiB back
find $O -mindepth 1 -maxdepth 1 ! -name r06f -exec rm -rf {} +
ls $K; for f in $K/r06_bench.json $K/r06_*_bench.json; do [ -f $f ] && tail -c 400 $f; echo; done
