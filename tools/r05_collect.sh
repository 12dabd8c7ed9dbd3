#!/bin/bash
# copies what tools/r05_final.sh (+ the per-config bench lines) left in gpurun_out/ into profiles/ under the round's names
cd "$(dirname "$0")/.." && cd gpurun_out || exit 1
cp r05f_bench.json ../profiles/r05_bench.json
cp r05f_bench_kernel_stats.csv ../profiles/r05_bench_kernel_stats.csv
cp pmc_r05f_cfg2_traffic.json ../profiles/r05_pmc_traffic.json
for k in k_iter_spec k_accum_tiles k_de_dir1 k_de_dir4 k_de_dir5 k_de_dir6; do cp r05f_sq_counters_$k.json ../profiles/r05_sq_counters_$k.json; done
for c in cfg3 cfg4 cfg5; do
  cp r05f_${c}_kernel_stats.csv ../profiles/r05_${c}_kernel_stats.csv
  cp pmc_r05f_${c}_traffic.json ../profiles/r05_${c}_pmc_traffic.json
  cp sq_r05f_${c}_iter.json ../profiles/r05_${c}_sq_counters_k_iter_spec.json
  cp sq_r05f_${c}_accum.json ../profiles/r05_${c}_sq_counters_k_accum_tiles.json
  [ -f r05f_${c}_bench.json ] && cp r05f_${c}_bench.json ../profiles/r05_${c}_bench.json
done
cp r05_de_slot_budget.txt ../profiles/r05_de_slot_budget.txt
