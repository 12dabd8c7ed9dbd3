#!/bin/bash
# Instruction-mix / stall counters of the bench kernels, a few SQ counters per pass.
# Usage: [BENCH_ARGS="--config cfg5"] tools/pmc_sq.sh <tag> [kernel-substring]
export TMPDIR=/tmp
export FLAME_LANES=1
tag=$1; kern=${2:-k_iter}
groups=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS_F32" \
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU" \
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC" "SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32")
i=0
for g in "${groups[@]}"; do
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d gpurun_out/sq_${tag}_$i -o b -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --preheat-seconds 0 --min-timed-frames 0 $BENCH_ARGS > gpurun_out/sq_${tag}_$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, collections, glob, json
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/sq_${tag}_*/b_counter_collection.csv")):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "${kern}" in r["Kernel_Name"]:
            acc[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (c, d), v in acc.items():
        per[c][d] = v
out = {}
for c, d in per.items():
    vals = sorted(d.values())
    out[c] = vals[len(vals) // 2]
import hashlib, os
lib = os.environ.get("FLAME_HIP_LIB", "cuburn_amd/_lib/libflame_hip.so")
out["_lib_sha256"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()          # bench.py quotes these counters only for the library they were measured on
json.dump(out, open("gpurun_out/sq_${tag}.json", "w"), indent=1, sort_keys=True)
for c in sorted(out):
    if not c.startswith("_"): print(c.ljust(28), "%.4g" % out[c])
PY
