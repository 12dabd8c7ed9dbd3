#!/usr/bin/env python3
"""Throughput of the device radix sort, the reference harness's numbers (cuburn/code/sort.py:524-546:
msec, Mkeys/s per pass) plus the HBM figure: 12 algorithmic bytes per key and pass.
    python tools/sort_bench.py [log2_count=25]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from cuburn_amd import render, _lib
from cuburn_amd.sort import Sorter

count = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 25)
fb = render.Framebuffers(device=0, nslots=1024, host_seed=1)
lib = _lib.load()
np.random.seed(42)
for bits in (7, 8, 9, 10):
    keys = np.uint32(np.random.randint(0, 1 << 32, size=count, dtype=np.uint64))
    src = torch.from_numpy(keys.view(np.int32)).to('cuda:0')
    a, b = torch.empty_like(src), torch.empty_like(src)
    s = Sorter(count, fb=fb); s.radix_bits = bits
    for _ in range(3):
        s.sort(a, src, count)
    lib.fl_ctx_sync(fb.ctx)
    t0 = time.perf_counter(); trials = 20
    for _ in range(trials):
        s.sort(a, src, count)
    lib.fl_ctx_sync(fb.ctx)
    ms = (time.perf_counter() - t0) / trials * 1e3
    print('%2d-bit pass over 2^%d keys: %.3f ms  %.0f Mkeys/s  %.0f GB/s of 12 B/key' % (bits, int(np.log2(count)), ms, count / ms / 1e3, 12.0 * count / ms / 1e6))
rounds = 4
s = Sorter(count, fb=fb)
out = s.multisort(a, b, src, count, rounds=rounds); lib.fl_ctx_sync(fb.ctx)
t0 = time.perf_counter()
for _ in range(5):
    out = s.multisort(a, b, src, count, rounds=rounds)
lib.fl_ctx_sync(fb.ctx)
ms = (time.perf_counter() - t0) / 5 * 1e3
ok = np.array_equal(out.cpu().numpy().view(np.uint32), np.sort(keys))
print('full 32-bit sort (4 x 8 bits): %.3f ms  %.0f Mkeys/s  correct=%s' % (ms, count / ms / 1e3, ok))
