#!/usr/bin/env python3
"""Where does the full-size filter chain differ most from the oracle?  Runs a BASELINE config's frame,
then the chain in two stages (yuv + bilateral | logscale + colorclip) on device and oracle, and reports
the error by stage and by density class, and the location of the maximum.
    python tools/diag_chain_error.py [cfg2|cfg3]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from common import O
from cuburn_amd import configs, render
import test_gpu_fullsize as T

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
gnm, prof = configs.CONFIGS[cfg]()
m = render.RenderManager(device=0, host_seed=42)
rdr, gprof, dim, td, nrun, front = T.iterate_frame(m, gnm, prof, 0.5, 2 ** 28)
d = O.calc_dim(gprof.width, gprof.height)
vals = {}
stages = {}
for filt in rdr.filts:
    vals[filt.name] = [float(v) for v in filt.scalars(gprof, getattr(gprof.filters, filt.name), dim, 0.5)]
    filt.apply(m.fb, gprof, getattr(gprof.filters, filt.name), dim, 0.5)
    if filt.name in ('bilateral', 'colorclip'):
        stages[filt.name] = m.fb.read('front', (dim.ah * dim.astride, 4), np.float32)
print('scalars', vals)
ref_de = O.bilateral_chain(d, O.yuv_to_rgb(d, np.ascontiguousarray(front)), *vals['bilateral'])
ref_out = O.colorclip(d, O.logscale(d, ref_de, *vals['logscale']), *vals['colorclip'])
acc_w = front[:, 3]
for name, dev, ref in (('DE output (linear)', stages['bilateral'], ref_de), ('tone-mapped', stages['colorclip'], ref_out)):
    err = np.abs(dev - ref)
    i = int(np.argmax(err.max(1)))
    y, x = divmod(i, dim.astride)
    print('%s: max abs %.3e at (x %d, y %d) accumulator density %.1f, DE density ref %.4e dev %.4e, values ref %s dev %s' % (
        name, err.max(), x, y, acc_w[i], ref_de[i, 3], stages['bilateral'][i, 3], ref[i], dev[i]))
    rel = err / np.maximum(np.abs(ref), 1e-30)
    w = ref_de[:, 3]
    for lo, hi in ((0, 1e-3), (1e-3, 1e-2), (1e-2, 0.1), (0.1, 1), (1, 10), (10, 1e9)):
        sel = (w >= lo) & (w < hi)
        if sel.any():
            print('   DE density [%g, %g): %8d px  max abs %.2e  max rel(w) %.2e  p99.9 abs %.2e' % (
                lo, hi, sel.sum(), err[sel].max(), rel[sel][:, 3].max(), np.percentile(err[sel], 99.9)))
m.fb.free()
