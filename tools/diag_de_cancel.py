#!/usr/bin/env python3
"""Where the DE differs from the oracle for a soak case of tools/soak_filters.py (round 5: the expanded colour term).
    python tools/diag_de_cancel.py <case> [<case> ...]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from common import O
from cuburn_amd import render, _lib
import test_gpu_parity as P
lib = _lib.load()
m = render.RenderManager(device=0, nslots=1024, host_seed=7)
for k in [int(v) for v in sys.argv[1:]]:
    rs = np.random.RandomState(7000 + k)
    w, h = int(rs.choice([96, 161, 320, 480, 641])), int(rs.choice([64, 97, 180, 270, 359]))
    dim = m.fb.set_dim(w, h); d = O.calc_dim(w, h)
    acc = (P.synth_accum if k % 2 == 0 else P.sparse_accum)(dim, seed=k + 1)
    buf = O.yuv_to_rgb(d, acc)
    bil = [float(rs.uniform(0.5, 12.0)), float(10 ** rs.uniform(-2.5, -0.3)), float(rs.uniform(0.3, 4.0)), float(rs.uniform(0.3, 1.2)), float(rs.uniform(0.5, 8.0))]
    _lib.check(lib.fl_debug_clear(m.fb.ctx, dim.w, dim.h, 0))
    m.fb.write('front', buf)
    arr = np.asarray(bil, np.float32)
    _lib.check(lib.fl_filter(m.fb.ctx, _lib.FILT['bilateral'], dim.w, dim.h, arr.ctypes.data, len(arr)))
    dev = m.fb.read('front', buf.shape, np.float32).reshape(dim.ah, dim.astride, 4)
    ref = O.bilateral_chain(d, buf, *bil).reshape(dim.ah, dim.astride, 4)
    b = buf.reshape(dim.ah, dim.astride, 4)
    with np.errstate(all='ignore'):
        n_in = np.where(b[..., 3:] > 0, b[..., :3] / b[..., 3:], 0)
        rel = np.abs(dev - ref) / (np.abs(ref) + 1e-3)
    iy, ix, ic = np.unravel_index(np.nanargmax(rel), rel.shape)
    print('case %d %dx%d bil=%s: input |n| max %.3g, p99 %.3g; worst rel %.3g at (%d,%d,ch %d): dev %s ref %s in %s' % (
        k, w, h, np.round(bil, 4), np.abs(n_in).max(), np.percentile(np.abs(n_in), 99), rel[iy, ix, ic], iy, ix, ic, dev[iy, ix], ref[iy, ix], b[iy, ix]))
    print('   count of values with rel > 1e-2: %d of %d; > 2e-3: %d' % ((rel > 1e-2).sum(), rel.size, (rel > 2e-3).sum()))
    ys, xs = slice(max(0, iy - 2), iy + 3), slice(max(0, ix - 2), ix + 3)
    print('   input w around:\n', np.round(b[ys, xs, 3], 3))
    print('   input n.x around:\n', np.round(n_in[ys, xs, 0], 3))
    # where do the deviations sit?  by magnitude of the oracle's density and by the input's emptiness
    w_ref = ref[..., 3]; bad = (rel[..., 3] > 2e-3)
    for lo, hi in ((0, 1e-12), (1e-12, 1e-6), (1e-6, 1e-3), (1e-3, 1e-1), (1e-1, 10), (10, 1e9)):
        sel = (w_ref >= lo) & (w_ref < hi)
        print('   oracle density in [%g, %g): %7d pixels, %6d off by > 2e-3 rel; of those input-empty: %d' % (lo, hi, sel.sum(), (sel & bad).sum(), (sel & bad & (b[..., 3] == 0)).sum()))
    # (round 5 ran the literal per-tap kernel and round 1's split form here as well — FLAME_DE_REFERENCE_FORM agreed with the oracle,
    # FLAME_DE_SPLIT deviated exactly like the default form — before both were removed from the library)
