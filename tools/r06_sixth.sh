#!/bin/bash
# Round 6: the batch sort's form for full batches (k_iter_spec, through hipRTC flags), and the chain's TCC traffic with the packed log
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for rep in 1 2; do for f in "-DFL_FULL_BATCH=0" "-DFL_FULL_BATCH=1"; do
  export FLAME_RTC_FLAGS="$f"
  echo "== $f"; tools/prof_kernels.sh fb_$rep --preheat-seconds 1.5 | grep -E "k_iter_spec"
  python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_frame'])"
done; done 2>&1 | tee gpurun_out/r06_sixth_fullbatch.txt
unset FLAME_RTC_FLAGS
tools/pmc_traffic.sh r06a > gpurun_out/r06_sixth_traffic.txt 2>&1; cat gpurun_out/r06_sixth_traffic.txt
