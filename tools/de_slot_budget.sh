#!/bin/bash
# round 5: the DE's vector-ALU slot budget per direction and phase, from hardware counters of three builds of the library:
#   full    cuburn_amd/_lib/libflame_hip.so
#   stage   -DDE_X_STOP_AFTER=4  (a workgroup ends when its planes are staged: loads, blurs, tap terms)
#   taps    -DDE_X_TAPSONLY      (a workgroup runs the tap loop + epilogue on whatever its LDS holds)
# (the timing builds produce garbage pictures).  Durations: rocprofv3 --kernel-trace --stats; instructions: SQ counters, separate passes.
# usage: tools/de_slot_budget.sh [bench args]     -> gpurun_out/${DE_BUDGET_TAG:-r06}_de_slot_budget.txt (with the sha256 of the full library: bench.py quotes the budget for that library only)
export TMPDIR=/tmp FLAME_LANES=1
declare -A LIB=([full]=cuburn_amd/_lib/libflame_hip.so [stage]=cuburn_amd/_lib/libflame_hip_xstage.so [taps]=cuburn_amd/_lib/libflame_hip_xtaps.so)
python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 --min-timed-frames 0 "$@" > /dev/null 2>&1
for b in full stage taps; do
  export FLAME_HIP_LIB=$PWD/${LIB[$b]}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sb_${b}_t -o b -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 60 "$@" > gpurun_out/sb_${b}_t.log 2>&1
  i=0
  for g in "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
    rocprofv3 --kernel-trace --pmc $g --output-format csv -d gpurun_out/sb_${b}_c$i -o b -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --preheat-seconds 0 --min-timed-frames 0 "$@" > gpurun_out/sb_${b}_c$i.log 2>&1
    i=$((i+1))
  done
done
python3 tools/de_slot_budget.py gpurun_out > gpurun_out/${DE_BUDGET_TAG:-r06}_de_slot_budget.txt
cat gpurun_out/${DE_BUDGET_TAG:-r06}_de_slot_budget.txt
