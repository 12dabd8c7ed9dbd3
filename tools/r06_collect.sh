#!/bin/bash
# copies what tools/r06_final.sh left in gpurun_out/r06f/ into profiles/ (the files already carry their profiles/ names)
cd "$(dirname "$0")/.." || exit 1
for f in gpurun_out/r06f/r06_*; do
  case $f in *.err|*_traffic.txt|*_kernels.txt) continue;; esac
  cp $f profiles/
done
ls profiles | grep '^r06_'
