#!/bin/bash
# DE parity (all filter / chain / edge tests) + per-direction times of the default library
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -q -x -k "filter or bilateral or de_ or deferred or chain or band or size or default" 2>&1 | tail -4
tools/ab_de.sh "$@"
