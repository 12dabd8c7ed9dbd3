#!/usr/bin/env python3
"""Band walker vs tile form of the DE, one direction at a time (FLAME_DE_UNFUSED_ENDS + FLAME_DE_DIR_MASK):
where do they differ?   python tools/diag_de_band.py [w h [seg_rows]]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from common import O
from cuburn_amd import render
import test_gpu_parity as P

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 120)
seg = sys.argv[3] if len(sys.argv) > 3 else None
os.environ['FLAME_DE_UNFUSED_ENDS'] = '1'
for pat in range(1, 8):
    os.environ['FLAME_DE_DIR_MASK'] = str(1 << pat)
    outs = {}
    for form in ('tiles', 'band'):
        os.environ.pop('FLAME_DE_BAND', None); os.environ.pop('FLAME_DE_SEG_ROWS', None)
        if form == 'band':
            os.environ['FLAME_DE_BAND'] = '1'
            if seg: os.environ['FLAME_DE_SEG_ROWS'] = seg
        m = render.RenderManager(device=0, nslots=1024, host_seed=7)
        dim = m.fb.calc_dim(w, h); d = O.calc_dim(w, h)
        buf = O.yuv_to_rgb(d, P.synth_accum(dim, seed=21))
        outs[form] = P.run_filter(m, 'bilateral', dim, buf, [6.0, 0.05, 1.5, 0.8, 4.0]).reshape(dim.ah, dim.astride, 4)
        m.fb.free()
    a, b = outs['tiles'], outs['band']
    bad = (a.view(np.uint32) != b.view(np.uint32)).any(2)
    ys, xs = np.nonzero(bad)
    print('direction %d: %d differing pixels of %d' % (pat, bad.sum(), bad.size), end='')
    if bad.any():
        print('  rows %d..%d cols %d..%d  max abs %.3e' % (ys.min(), ys.max(), xs.min(), xs.max(), np.abs(a - b).max()))
        rows = np.bincount(ys, minlength=dim.ah)
        print('   differing pixels per row:', ' '.join('%d:%d' % (y, n) for y, n in enumerate(rows) if n))
        cols = np.bincount(xs, minlength=dim.astride)
        print('   per column:', ' '.join('%d:%d' % (x, n) for x, n in enumerate(cols) if n))
    else:
        print()
