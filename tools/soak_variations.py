#!/usr/bin/env python3
"""One-off soak: every variation with RANDOM parameters (several draws each, both signs, small and
large magnitudes) on points from a wider range than tests/test_gpu_variations.py uses, device against
the oracle, same acceptance rule (99 % of points within 2e-3 rel / 2e-4 abs, RNG draws bit-exact).
    python tools/soak_variations.py [draws=4]"""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from common import O, mwc
from cuburn_amd import configs, profile, render, _lib
from cuburn_amd.genome import variations as V

draws = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = 2048
lib = _lib.load()
m = render.RenderManager(device=0, nslots=1024, host_seed=7)
L = O.lib()
L.ref_apply_xf.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
bad = []
for name in sorted(V.var_ids, key=lambda n: V.var_ids[n]):
    for k in range(draws):
        rs = np.random.RandomState(1000 * V.var_ids[name] + k)
        params = {}
        for pn, (dv, _) in V.var_params[name].items():
            base = dv if dv else 0.6
            params[pn] = float(base * rs.choice([-1.0, 1.0]) * rs.choice([0.2, 0.7, 1.0, 1.9, 3.3])) if k else float(base)
        params['weight'] = float(rs.choice([-0.7, 0.35, 0.8, 1.6]))
        gnm = {'type': 'animation', 'camera': {'scale': 0.25}, 'time': {'duration': 1, 'frame_width': 0.0},
               'palette': [[0.0] + configs.palette_encode(configs.grey_ramp())],
               'xforms': {'0': {'weight': 1.0, 'color': 0.7, 'color_speed': 0.3,
                                'pre_affine': configs._affine(float(rs.uniform(-180, 180)), float(rs.uniform(0.3, 1.8)), float(rs.uniform(-0.5, 0.5)), float(rs.uniform(-0.5, 0.5))),
                                'variations': {name: params}}}}
        prof = {'width': 64, 'height': 64, 'spp': 1, 'fps': 1, 'duration': 1, 'frame_width': 0}
        gprof = profile.wrap(prof, gnm)
        rdr = render.Renderer(gnm, gprof)
        g = rdr._handle(m.fb); m._copy(rdr, gnm)
        _lib.check(lib.fl_interp(m.fb.ctx, g, 64, 64, 0.5, 0.0))
        P = m.fb.read('params', (m.fb.nslots, rdr.packer.pstride), np.float32, g)[5]
        pts = np.zeros((N, 4), np.float32)
        span = float(rs.choice([0.3, 1.5, 5.0]))
        pts[:, 0] = rs.uniform(-span, span, N); pts[:, 1] = rs.uniform(-span, span, N); pts[:, 2] = rs.uniform(0, 1, N)
        rng = mwc.make_seeds(N, 99 + k)
        dev_pts, dev_rng = pts.copy(), rng.copy()
        _lib.check(lib.fl_debug_apply_xf(m.fb.ctx, g, 5, 0, N, dev_pts.ctypes.data, dev_rng.ctypes.data))
        ref_pts, ref_rng = pts.copy(), rng.copy()
        for i in range(N):
            x, y, c = C.c_float(pts[i, 0]), C.c_float(pts[i, 1]), C.c_float(pts[i, 2])
            st = ref_rng[i:i + 1]
            L.ref_apply_xf(rdr.packer.prog.ctypes.data, P.ctypes.data, 0, C.byref(x), C.byref(y), C.byref(c), st.ctypes.data)
            ref_pts[i, :3] = (x.value, y.value, c.value)
        d, r = dev_pts[:, :2].astype(np.float64), ref_pts[:, :2].astype(np.float64)
        fin = np.isfinite(r).all(1) & (np.abs(r).max(1) < 1e6)
        both_bad = ~np.isfinite(d).all(1) & ~np.isfinite(r).all(1)
        ok = np.zeros(N, bool)
        ok[fin] = (np.abs(d[fin] - r[fin]) <= 2e-4 + 2e-3 * np.abs(r[fin])).all(1)
        ok |= both_bad | (~fin & ~both_bad & (np.abs(r).max(1) >= 1e6))
        rng_ok = np.array_equal(dev_rng, ref_rng)
        if ok.mean() < 0.99 or not rng_ok:
            bad.append((name, k, round(float(ok.mean()), 4), rng_ok, params, span))
            print('FAIL', bad[-1], d[~ok][:2], r[~ok][:2], flush=True)
print('%d variation x draw cases, %d failures' % (len(V.var_ids) * draws, len(bad)))
