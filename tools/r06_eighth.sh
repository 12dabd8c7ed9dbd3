#!/bin/bash
# Round 6: directory blocks (FLAME_DIR_GROUP = log2 G) on every bench config, against the [tile][batch] directory of commit 3ccf514
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --preheat-seconds 2 > /dev/null 2>&1
run() { python bench.py --config $2 --steps $3 --warmup 1 --cpu-seconds 0 --preheat-seconds 1 --min-timed-frames $4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2 $1'.ljust(12), d['value'], d['ms_per_step'], d['kernel_ms_per_frame'], d['roofline']['frac'])"; }
for cfg in cfg2 cfg3 cfg4 cfg5; do
  case $cfg in cfg2) st=8; mf=300;; cfg3) st=6; mf=100;; cfg4) st=6; mf=120;; cfg5) st=4; mf=16;; esac
  for rep in 1 2; do
    FLAME_HIP_LIB=$PWD/cuburn_amd/_lib/libflame_hip_c1.so run c1 $cfg $st $mf
    for g in 1 2; do FLAME_DIR_GROUP=$g run g$g $cfg $st $mf; done
  done
done 2>&1 | tee gpurun_out/r06_eighth_ab.txt
