#!/bin/bash
# Cache policy of the accumulate's record loads (library variants built with -DACC_LOAD_MOD='" nt"' etc.) and of the
# iterate kernel's log stores (FLAME_RTC_FLAGS=-DFL_LOG_NT=1): rocprofv3 averages of k_iter_spec and k_accum_tiles, one box
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
for v in "libflame_hip.so:" "libflame_hip_nt.so:" "libflame_hip_sc1.so:" "libflame_hip_sc0.so:" "libflame_hip.so:-DFL_LOG_NT=1" "libflame_hip_nt.so:-DFL_LOG_NT=1" "libflame_hip.so:"; do
  L=${v%%:*}; F=${v#*:}
  export FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/$L FLAME_RTC_FLAGS="$F"
  echo "== $L [$F]"; tools/prof_kernels.sh cp_${L%.so}_${#F} --preheat-seconds 1.0 2>&1 | grep -E "k_iter|k_accum" | cut -c1-20,64-100
done
