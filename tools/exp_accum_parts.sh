#!/bin/bash
# what bounds k_accum_tiles: builds with the log load / palette load / LDS atomic knocked out (timing only, wrong results)
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for L in libflame_hip.so libflame_hip_NOLOG.so libflame_hip_NOPAL.so libflame_hip_NOATOM.so libflame_hip.so; do
  FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/$L python bench.py --config ${1:-cfg2} --steps 6 --warmup 1 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${1:-cfg2}', '$L'.ljust(26), d['ms_per_step'], d['kernel_ms_per_frame'])"
done
