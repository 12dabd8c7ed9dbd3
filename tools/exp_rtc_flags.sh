#!/bin/bash
# k_iter_spec under extra hipRTC code-generation options (FLAME_RTC_FLAGS): rocprofv3 average per variant, one box.
# A variant whose option the compiler rejects falls back to the interpreter kernel (then "k_iter<" shows up instead).
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
i=0
while IFS= read -r flags; do
  export FLAME_RTC_FLAGS="$flags"
  echo "== [$flags]"; tools/prof_kernels.sh rtcf_$i --preheat-seconds 1.0 2>&1 | grep -E "k_iter" | head -2
  i=$((i+1))
done <<'LIST'

-O2
-mllvm -amdgpu-skip-threshold=4
-mllvm -amdgpu-skip-threshold=40
-mllvm -enable-post-misched=false
-mllvm -amdgpu-sched-strategy=max-ilp
-mllvm -amdgpu-sched-strategy=iterative-ilp
-mllvm -amdgpu-sched-strategy=iterative-minreg
-mllvm -amdgpu-schedule-metric-bias=0
-mllvm -amdgpu-igrouplp=false
-mllvm -amdgpu-disable-unclustered-high-rp-reschedule=true
-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=true
LIST
