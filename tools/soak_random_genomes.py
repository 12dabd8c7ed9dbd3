#!/usr/bin/env python3
"""Extended run of tests/test_gpu_random_genomes.py::test_random_genome_parity over seeds outside the committed 48
(default schedule: fuse 256).   python tools/soak_random_genomes.py [first=49] [last=200]"""
import os, sys, traceback
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cuburn_amd import render
import test_gpu_random_genomes as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 49
last = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mgr = render.RenderManager(device=0, nslots=T.NSLOTS, host_seed=23)
mgr_prod = render.RenderManager(device=0, host_seed=23)
parity = getattr(T.test_random_genome_parity, '__wrapped__', T.test_random_genome_parity)
bad = []
for seed in range(first, last + 1):
    try:
        parity(mgr, mgr_prod, seed)
        print('ok   seed %d (%d xforms)' % (seed, len(T.random_genome(seed)[0]['xforms'])), flush=True)
    except AssertionError as e:
        bad.append(seed)
        print('FAIL seed %d (%d xforms): %s' % (seed, len(T.random_genome(seed)[0]['xforms']), str(e)[:200]), flush=True)
print('%d seeds, %d failures: %s' % (last - first + 1, len(bad), bad))
