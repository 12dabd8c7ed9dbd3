#!/usr/bin/env python3
"""One-off soak: every output pixel format at random image sizes with inputs that include negatives,
values above 1, zeros, NaN and infinities, device against the oracle, bit for bit incl. RNG states."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from common import O
from cuburn_amd import render, _lib

lib = _lib.load()
m = render.RenderManager(device=0, nslots=1024, host_seed=5)
nwalk = 1024 * 256
nbad = 0
for k in range(36):
    rs = np.random.RandomState(k)
    w, h = int(rs.randint(1, 400)) * 2, int(rs.randint(1, 300)) * 2
    if k % 6 == 5:
        w += 1                                  # odd width: every format but 4:2:0
    dim = m.fb.set_dim(w, h); d = O.calc_dim(w, h)
    buf = rs.uniform(-0.3, 1.4, (dim.ah * dim.astride, 4)).astype(np.float32)
    buf[rs.uniform(size=len(buf)) < 0.2] = 0.0
    sel = rs.uniform(size=buf.shape)
    buf[sel < 0.002] = np.nan; buf[(sel >= 0.002) & (sel < 0.004)] = np.inf; buf[(sel >= 0.004) & (sel < 0.006)] = -np.inf
    for fmt in range(6):
        if fmt == 4 and (w % 2 or h % 2):
            continue
        _lib.check(lib.fl_debug_clear(m.fb.ctx, dim.w, dim.h, 0))
        m.fb.write('front', buf)
        seeds = m.fb.read('seeds', (m.fb.nwalkers, 3), np.uint32)
        ref, rng_after = O.f32_to_rgba(d, buf, seeds[nwalk + 64 * 256:], fmt)
        out = np.zeros_like(ref)
        _lib.check(lib.fl_output(m.fb.ctx, w, h, fmt, out.ctypes.data, 0))
        _lib.check(lib.fl_ctx_sync(m.fb.ctx))
        after = m.fb.read('seeds', (m.fb.nwalkers, 3), np.uint32)
        ok = np.array_equal(out, ref) and np.array_equal(after[nwalk + 64 * 256:], rng_after)
        if not ok:
            nbad += 1
            print('FAIL %dx%d fmt %d: %d differing values' % (w, h, fmt, int((out != ref).sum())), flush=True)
print('%d failures' % nbad)
