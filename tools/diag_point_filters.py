#!/usr/bin/env python3
"""One-off: the full filter chain on accumulators with a point attractor (one pixel holding 30-100 %
of all samples next to empty / sparse surroundings), device against the oracle's chain."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from common import O
from cuburn_amd import render
import test_gpu_parity as P
import test_gpu_fullsize as F

for share in (1.0, 0.9, 0.6, 0.3, 0.03):
    gnm, prof = P.point_flame(share)
    prof = dict(prof, filter_order=['bilateral', 'logscale', 'colorclip'])
    m = render.RenderManager(device=0, host_seed=42)
    rdr, gprof, dim, td, nrun, front = F.iterate_frame(m, gnm, prof, 0.5, 2 ** 26)
    vals, dev = F.filter_chain_on_device(m, rdr, gprof, dim, 0.5)
    d = O.calc_dim(gprof.width, gprof.height)
    ref = F.oracle_chain(d, front, vals)
    err = np.abs(dev - ref)
    print('share %.2f: max density %.3g of %.3g; chain err max %.3g mean %.3g p99.9 %.3g; finite %s; out max %.3f' % (
        share, front[:, 3].max(), front[:, 3].sum(), err.max(), err.mean(), np.percentile(err, 99.9), np.isfinite(dev).all(), dev.max()))
    m.fb.free()
