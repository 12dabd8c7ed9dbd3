#!/bin/bash
# round 5, first call: the MFMA tap micro-benchmark; SQ counters of the DE directions 4 and 6 (and 5 for comparison)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/mfma_tap_bench > gpurun_out/r05_mfma_tap_bench.txt 2>&1
tools/mfma_tap_bench >> gpurun_out/r05_mfma_tap_bench.txt 2>&1
cat gpurun_out/r05_mfma_tap_bench.txt
for p in 4 6 5; do
  tools/pmc_sq.sh r05_de$p "k_de_dir<$p" > gpurun_out/r05_sq_de$p.txt 2>&1
  cp gpurun_out/sq_r05_de$p.json gpurun_out/r05_sq_counters_k_de_dir$p.json
  echo "== dir $p"; cat gpurun_out/r05_sq_de$p.txt | tail -26
done
