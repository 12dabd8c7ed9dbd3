#!/usr/bin/env python3
"""(debug, round 5) the DE after k directions against k oracle passes for a soak case of tools/soak_filters.py: where the chain first
leaves the oracle (--detail / --save / --probe).  Needs a library with a three-line debug hook that is NOT in the tree: in
fl_filter's FLAME_DE_UNFUSED_ENDS branch (round-5 commit 978bba7's flame_abi.hip) bound the direction loop by
`getenv("FLAME_DE_NDIRS") ? atoi(getenv("FLAME_DE_NDIRS")) : 8`.  What it found is written up in
tests/test_gpu_parity.py::test_filter_bilateral_underflow_frontier and DESIGN.md 4.3."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from common import O
from cuburn_amd import render, _lib
import test_gpu_parity as P
lib = _lib.load()
os.environ['FLAME_DE_UNFUSED_ENDS'] = '1'
k = int(sys.argv[1])
rs = np.random.RandomState(7000 + k)
w, h = int(rs.choice([96, 161, 320, 480, 641])), int(rs.choice([64, 97, 180, 270, 359]))
d = O.calc_dim(w, h)
bil = None
ref = None
for nd in range(1, 9):
    os.environ['FLAME_DE_NDIRS'] = str(nd)
    m = render.RenderManager(device=0, nslots=1024, host_seed=7)
    dim = m.fb.set_dim(w, h)
    if bil is None:
        acc = (P.synth_accum if k % 2 == 0 else P.sparse_accum)(dim, seed=k + 1)
        buf = O.yuv_to_rgb(d, acc)
        bil = [float(rs.uniform(0.5, 12.0)), float(10 ** rs.uniform(-2.5, -0.3)), float(rs.uniform(0.3, 4.0)), float(rs.uniform(0.3, 1.2)), float(rs.uniform(0.5, 8.0))]
        ref = buf.copy()
    _lib.check(lib.fl_debug_clear(m.fb.ctx, dim.w, dim.h, 0))
    m.fb.write('front', buf)
    arr = np.asarray(bil, np.float32)
    _lib.check(lib.fl_filter(m.fb.ctx, _lib.FILT['bilateral'], dim.w, dim.h, arr.ctypes.data, len(arr)))
    dev = m.fb.read('front', buf.shape, np.float32).reshape(dim.ah, dim.astride, 4)
    prev = ref.copy()
    ref, b1, b2 = O.bilateral_pass(d, ref, nd - 1, bil[0] * w / 1920.0 if False else bil[0], *bil[1:])
    r = ref.reshape(dim.ah, dim.astride, 4)
    with np.errstate(all='ignore'):
        rel = np.abs(dev - r) / (np.abs(r) + 1e-3 * 0 + 1e-30)
        rel[~np.isfinite(rel)] = 0
    sig = (np.abs(dev - r) > 2e-4 + 2e-3 * np.abs(r))
    iy, ix, ic = np.unravel_index(np.argmax(np.where(sig, np.abs(dev - r), 0)), rel.shape)
    print('after %d directions: %6d values outside 2e-3 rel + 2e-4 abs; largest abs diff %.3g at (%d,%d,ch%d): dev %s ref %s' % (nd, sig.sum(), np.abs(dev - r)[iy, ix, ic], iy, ix, ic, dev[iy, ix], r[iy, ix]))
    if '--probe' in sys.argv:
        py, px = 50, 391
        print('   probe (%d,%d): dev %s ref %s' % (py, px, dev[py, px], r[py, px]))
        print('   probe row dev w:', dev[py, px - 3:px + 4, 3], ' ref w:', r[py, px - 3:px + 4, 3])
    if sig.sum() and '--save' in sys.argv:
        np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'de_cancel_case%d_pass%d.npz' % (k, nd)), prev=prev.reshape(dim.ah, dim.astride, 4), dev=dev, ref=r, b2=b2.reshape(dim.ah, dim.astride), bil=np.array(bil), pos=np.array([iy, ix]))
        break
    if sig.sum() and '--detail' in sys.argv:
        pv = prev.reshape(dim.ah, dim.astride, 4)
        print('   input of this pass around it (w):\n', pv[max(0, iy - 3):iy + 4, max(0, ix - 3):ix + 4, 3])
        print('   input of this pass around it (x/w):\n', pv[max(0, iy - 3):iy + 4, max(0, ix - 3):ix + 4, 0] / np.maximum(pv[max(0, iy - 3):iy + 4, max(0, ix - 3):ix + 4, 3], 1e-45))
        break
    m.fb.free()
