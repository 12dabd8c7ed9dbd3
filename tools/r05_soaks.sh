#!/bin/bash
# round 5: the soak tools against the round's final build (seeds and cases OUTSIDE the committed tests) -> gpurun_out/r05_soaks.txt
O=gpurun_out/r05_soaks.txt
echo "# Round 5: the soak tools against the round's final build (tools/soak_*.py; seeds and cases OUTSIDE the committed tests)" > $O
for t in "soak_random_genomes.py 49 140" "soak_filters.py 120" "soak_filters2.py 40" "soak_interp.py" "soak_output.py" "soak_slots.py" "soak_variations.py"; do
  echo "=== $t" >> $O
  timeout 1500 python tools/$t 2>&1 | grep -v "amdgpu.ids" | grep -v "^ok " | tail -12 >> $O
done
cat $O
