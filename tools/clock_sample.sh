#!/bin/bash
# sample the shader clock while bench.py runs
( for i in $(seq 1 40); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1; sleep 0.5; done ) > gpurun_out/r04_clocks.txt 2>&1 &
python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0 --preheat-seconds 3 > gpurun_out/r04_clk_bench.json 2>/dev/null
wait
sort gpurun_out/r04_clocks.txt | uniq -c | sort -rn | head -20
