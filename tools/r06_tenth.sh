#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_parity.py tests/test_gpu_bench.py tests/test_gpu_edges.py -x -q -m gpu -k "rccl or sharded or bench or geometry or long_launch" > gpurun_out/r06_tenth_tests.txt 2>&1
tail -15 gpurun_out/r06_tenth_tests.txt
