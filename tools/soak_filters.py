#!/usr/bin/env python3
"""One-off soak: the bilateral chain (+ logscale + colorclip) with RANDOM scalar parameters on dense and
sparse synthetic accumulators and odd image sizes, device against the oracle.
    python tools/soak_filters.py [cases=40]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from common import O
from cuburn_amd import render, _lib
import test_gpu_parity as P

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lib = _lib.load()
m = render.RenderManager(device=0, nslots=1024, host_seed=7)
worst = []
only = [int(v) for v in os.environ['SOAK_ONLY'].split(',')] if os.environ.get('SOAK_ONLY') else None
for k in (only or range(cases)):
    rs = np.random.RandomState(7000 + k)
    w, h = int(rs.choice([96, 161, 320, 480, 641])), int(rs.choice([64, 97, 180, 270, 359]))
    dim = m.fb.set_dim(w, h); d = O.calc_dim(w, h)
    acc = (P.synth_accum if k % 2 == 0 else P.sparse_accum)(dim, seed=k + 1)
    buf = O.yuv_to_rgb(d, acc)
    bil = [float(rs.uniform(0.5, 12.0)), float(10 ** rs.uniform(-2.5, -0.3)), float(rs.uniform(0.3, 4.0)), float(rs.uniform(0.3, 1.2)), float(rs.uniform(0.5, 8.0))]
    log = [float(rs.uniform(1.0, 8.0)), float(10 ** rs.uniform(-4, -1.5))]
    gam = float(rs.uniform(0.15, 0.6)); lin = float(10 ** rs.uniform(-3, -1))
    clip = [float(rs.uniform(0.0, 1.0)), float(rs.choice([-1.0, -0.5, 0.0, 0.5, 2.0])), gam, lin, lin ** (gam - 1.0)]
    _lib.check(lib.fl_debug_clear(m.fb.ctx, dim.w, dim.h, 0))
    m.fb.write('front', buf)
    for name, vals in (('bilateral', bil), ('logscale', log), ('colorclip', clip)):
        arr = np.asarray(vals, np.float32)
        _lib.check(lib.fl_filter(m.fb.ctx, _lib.FILT[name], dim.w, dim.h, arr.ctypes.data, len(arr)))
    dev = m.fb.read('front', buf.shape, np.float32)
    ref = O.colorclip(d, O.logscale(d, O.bilateral_chain(d, buf, *bil), *log), *clip)
    err = np.abs(dev - ref)
    bad_fin = int((~np.isfinite(dev) & np.isfinite(ref)).sum())
    worst.append((float(err[np.isfinite(err)].max()), k))
    flag = 'FAIL' if (bad_fin or err[np.isfinite(err)].max() > 3e-2 or np.percentile(err[np.isfinite(err)], 99.9) > 3e-3) else 'ok'
    print('%s case %2d %dx%d %s: max %.2e p99.9 %.2e mean %.2e nonfinite-on-device %d  bil=%s' % (flag, k, w, h, 'dense' if k % 2 == 0 else 'sparse',
          err[np.isfinite(err)].max(), np.percentile(err[np.isfinite(err)], 99.9), err[np.isfinite(err)].mean(), bad_fin, np.round(bil, 3)), flush=True)
print('worst', sorted(worst)[-3:])
