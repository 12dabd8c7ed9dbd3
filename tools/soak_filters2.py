#!/usr/bin/env python3
"""One-off soak, second set: the DE with extreme scalars (negative gradient speed, tiny / huge spatial
and density deviations, density power 0..2) and every other filter with random scalars, device
against the oracle on dense and sparse accumulators.
    python tools/soak_filters2.py [cases=30]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from common import O
from cuburn_amd import render, _lib
import test_gpu_parity as P

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lib = _lib.load()
m = render.RenderManager(device=0, nslots=1024, host_seed=7)


def run(dim, buf, name, vals):
    _lib.check(lib.fl_debug_clear(m.fb.ctx, dim.w, dim.h, 0))
    m.fb.write('front', buf)
    arr = np.asarray(vals, np.float32)
    _lib.check(lib.fl_filter(m.fb.ctx, _lib.FILT[name], dim.w, dim.h, arr.ctypes.data if len(arr) else None, len(arr)))
    return m.fb.read('front', buf.shape, np.float32)


def report(tag, k, dev, ref, vals, rel=2e-3, ab=2e-4):
    fin = np.isfinite(ref)
    bad_fin = int((~np.isfinite(dev) & fin).sum())
    err = np.abs(dev - ref)[fin]
    viol = float((err > ab + rel * np.abs(ref[fin])).mean())
    flag = 'FAIL' if bad_fin or viol > 1e-3 else 'ok'
    print('%s %s case %2d: violations %.2e max %.2e nonfinite-on-device %d  vals=%s' % (flag, tag, k, viol, err.max() if err.size else 0, bad_fin, np.round(vals, 4)), flush=True)


only = [int(v) for v in os.environ['SOAK_ONLY'].split(',')] if os.environ.get('SOAK_ONLY') else None
for k in (only or range(cases)):
    rs = np.random.RandomState(9000 + k)
    w, h = int(rs.choice([161, 320, 480])), int(rs.choice([97, 180, 270]))
    dim = m.fb.set_dim(w, h); d = O.calc_dim(w, h)
    acc = (P.synth_accum if k % 2 == 0 else P.sparse_accum)(dim, seed=k + 1)
    buf = O.yuv_to_rgb(d, acc)
    bil = [float(10 ** rs.uniform(-1.3, 1.7)), float(10 ** rs.uniform(-1.7, -0.3)), float(10 ** rs.uniform(-1.3, 1.0)), float(rs.uniform(0.0, 2.0)), float(rs.uniform(-8.0, 8.0))]
    report('bilateral', k, run(dim, buf, 'bilateral', bil), O.bilateral_chain(d, buf, *bil), bil)
    report('yuv', k, run(dim, acc, 'yuv', []), O.yuv_to_rgb(d, acc), [], 1e-5, 1e-6)
    log = [float(rs.uniform(0.5, 10.0)), float(10 ** rs.uniform(-5, -1))]
    lg = O.logscale(d, buf, *log)
    report('logscale', k, run(dim, buf, 'logscale', log), lg, log, 1e-3, 1e-5)
    gam = float(rs.uniform(0.1, 0.9)); lin = float(10 ** rs.uniform(-3.5, -0.5))
    clip = [float(rs.uniform(0.0, 1.0)), float(rs.uniform(-2.0, 3.0)), gam, lin, lin ** (gam - 1.0)]
    report('colorclip', k, run(dim, lg, 'colorclip', clip), O.colorclip(d, lg, *clip), clip, 2e-3, 2e-5)
    sm = [float(rs.uniform(0.3, 2.0)), gam - 1.0, lin, lin ** (gam - 1.0)]
    report('smearclip', k, run(dim, lg, 'smearclip', sm), O.smearclip_chain(d, lg, *sm), sm, 2e-3, 2e-5)
    report('haloclip', k, run(dim, lg, 'haloclip', [gam - 1.0]), O.haloclip_chain(d, lg, gam - 1.0), [gam - 1.0], 2e-3, 2e-5)
    pc = [gam - 1.0, lin, lin ** (gam - 1.0), float(rs.uniform(0.5, 4.0))]
    report('plainclip', k, run(dim, lg, 'plainclip', pc), O.plainclip(d, lg, *pc), pc, 2e-3, 2e-5)
    pos = np.maximum(lg, 1e-4)
    report('logencode', k, run(dim, pos, 'logencode', [2.2]), O.logencode(d, pos, 2.2), [2.2], 1e-3, 1e-4)
