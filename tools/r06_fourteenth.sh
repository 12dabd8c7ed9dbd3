#!/bin/bash
# Round 6: the batch sort without its per-record tests (timing experiment through hipRTC flags; every batch of cfg2 is full)
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for rep in 1 2; do for f in "-DFL_X_FULL_ONLY=0" "-DFL_X_FULL_ONLY=1"; do
  export FLAME_RTC_FLAGS="$f"
  echo "== $f"; tools/prof_kernels.sh fo_$rep --preheat-seconds 1.5 | grep -E "k_iter_spec"
  python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_frame'])"
done; done 2>&1 | tee gpurun_out/r06_fourteenth.txt
