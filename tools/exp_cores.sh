#!/bin/bash
# Co-residency experiment: can the accumulate / filter kernels of frame k share the CUs with the
# iterate kernel of frame k+1?  (5 iterate workgroups per CU + one 32 KB accumulate tile.)
# usage: tools/exp_cores.sh OUTDIR
out=${1:-gpurun_out/exp_cores}; mkdir -p $out
run() { # name, lib, env...
  name=$1; lib=$2; shift 2
  env "$@" FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/$lib python bench.py --steps 40 --warmup 5 --cpu-seconds 0 > $out/$name.json 2> $out/$name.err
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d['kernel_ms_per_frame']
    print('%-28s value %9.1f Ms/s  ms/frame %.3f  [1 lane: iter %.3f accum %.3f filt %.3f]  nslots %s' % (
        sys.argv[2], d['value'], d['ms_per_step'], k['iter'], k['accum_flush'], k['filters'], d['config'].get('nslots')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run base            libflame_hip.so     X=1
run base_1280_r12   libflame_hip.so     FLAME_NSLOTS=1280 FLAME_BIN_ROUNDS=12
run h32             libflame_hip_h32.so X=1
run h32_1280_r12    libflame_hip_h32.so FLAME_NSLOTS=1280 FLAME_BIN_ROUNDS=12
run h32_1280_r16    libflame_hip_h32.so FLAME_NSLOTS=1280
run h32_1024_r12    libflame_hip_h32.so FLAME_NSLOTS=1024 FLAME_BIN_ROUNDS=12
run base_again      libflame_hip.so     X=1
