#!/bin/bash
# code-generation variants of k_accum_tiles (libflame_hip_v*.so built with BINNED_FLAGS=...): the kernel's time
# moves by tens of per cent with its code layout
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for L in cuburn_amd/_lib/libflame_hip.so cuburn_amd/_lib/libflame_hip_v*.so cuburn_amd/_lib/libflame_hip.so; do
  FLAME_HIP_LIB=$PWD/$L python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1].ljust(24), d['ms_per_step'], d['kernel_ms_per_frame'])"
done
