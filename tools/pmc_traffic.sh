#!/bin/bash
# HBM traffic of the bench kernels from the TCC counters, two separate passes (FETCH_SIZE needs 3
# TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage: tools/pmc_traffic.sh <tag>
export TMPDIR=/tmp
tag=$1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_${tag}_$ctr -o b -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --min-timed-frames 0 $BENCH_ARGS > gpurun_out/pmc_${tag}_$ctr.log 2>&1
done
python3 - <<PY
import csv, collections, json
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open("gpurun_out/pmc_${tag}_%s/b_counter_collection.csv" % ctr)))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        per[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, d in per.items():
        vals = sorted(d.values())
        out.setdefault(k, {})[ctr] = {"n": len(vals), "median_per_launch": vals[len(vals) // 2], "max": vals[-1],
                                      "mean_per_launch": sum(vals) / len(vals)}       # (a frame's launches differ in length: launches x mean = the frame)
import hashlib, os, subprocess
lib = os.environ.get("FLAME_HIP_LIB", "cuburn_amd/_lib/libflame_hip.so")
out["_meta"] = {"lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(), "lib": lib, "bench_args": os.environ.get("BENCH_ARGS", "")}
json.dump(out, open("gpurun_out/pmc_${tag}_traffic.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(((k, v) for k, v in out.items() if k != "_meta"), key=lambda kv: -kv[1].get("WRITE_SIZE", {}).get("median_per_launch", 0)):
    print(k[:60].ljust(60), {c: round(x["median_per_launch"], 1) for c, x in v.items()})
PY
