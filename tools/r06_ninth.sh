#!/bin/bash
# Round 6: after the clean-up of binned.hip (always-executed waits): binned parity tests, workgroups per tile with the lean accumulate
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "binned or bit_exact or larger_workgroups or cfg2 or attractor or pipelined or long_launch or hot or 8k or cfg5 or cfg4 or cfg3" > gpurun_out/r06_ninth_tests.txt 2>&1
tail -3 gpurun_out/r06_ninth_tests.txt
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for p in 16 8 12 24 16; do
  export FLAME_BIN_PARTS=$p
  echo "== parts $p"; tools/prof_kernels.sh parts_$p --preheat-seconds 1.5 | grep -E "k_accum"
  python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_frame'])"
done 2>&1 | tee gpurun_out/r06_ninth_parts.txt
