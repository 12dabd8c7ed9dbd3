"""
GPU parity of every variation: one application of an xform holding that variation (plus a
non-trivial pre and post affine) to 4096 points, HIP kernel vs oracle, through the C ABI.

Tolerance: the device uses single hardware instructions for sin/cos/exp/log/rcp/sqrt (as the
reference does under -use_fast_math, cuburn/code/util.py:96), so results agree to ~1e-4
relative for smooth variations: every point must lie within 2e-3 relative / 2e-4 absolute of the
oracle's, RNG streams advance identically (bit-exact), colour exact.  Variations with floor / trunc /
compare branches can flip a branch on a 1-ulp difference of an intermediate, and a few are ill-conditioned
in places (sin(tan(3y)) next to a pole of tan); a point outside the tolerance is accepted ONLY if the
device's result is no further from the oracle's than the oracle's own results spread over the inputs
within 6 ulp of the point (a discontinuity, or a condition number that makes 1 ulp of input worth more than the tolerance),
and at most 1.5 % of the points may need that — anything else fails (round 3 accepted any 1 % of points
unexamined).
"""
import ctypes as C

import numpy as np
import pytest

from common import O, prepare, mwc
from cuburn_amd import configs, profile, render, _lib
from cuburn_amd.genome import variations as V

pytestmark = pytest.mark.gpu
N = 4096


@pytest.fixture(scope='module')
def mgr(built):
    return render.RenderManager(device=0, nslots=1024, host_seed=7)


def genome_for(name):
    params = dict((k, dv if dv else 0.6) for k, (dv, _) in V.var_params[name].items())
    params['weight'] = 0.8
    return {
        'type': 'animation', 'camera': {'scale': 0.25}, 'time': {'duration': 1, 'frame_width': 0.0},
        'palette': [[0.0] + configs.palette_encode(configs.grey_ramp())],
        'xforms': {'0': {'weight': 1.0, 'color': 0.7, 'color_speed': 0.3,
                         'pre_affine': configs._affine(20.0, 0.9, 0.3, 0.2),
                         'post_affine': configs._affine(-10.0, 1.1, -0.1, 0.05),
                         'variations': {name: params}}},
    }


@pytest.mark.parametrize('name', sorted(V.var_ids, key=lambda n: V.var_ids[n]))
def test_variation_matches_oracle(mgr, name):
    lib = _lib.load()
    gnm = genome_for(name)
    prof = {'width': 64, 'height': 64, 'spp': 1, 'fps': 1, 'duration': 1, 'frame_width': 0}
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    g = rdr._handle(mgr.fb)
    mgr._copy(rdr, gnm)
    _lib.check(lib.fl_interp(mgr.fb.ctx, g, 64, 64, 0.5, 0.0))
    params = mgr.fb.read('params', (mgr.fb.nslots, rdr.packer.pstride), np.float32, g)

    rs = np.random.RandomState(V.var_ids[name])
    pts = np.zeros((N, 4), np.float32)
    pts[:, 0] = rs.uniform(-1.5, 1.5, N)
    pts[:, 1] = rs.uniform(-1.5, 1.5, N)
    pts[:, 2] = rs.uniform(0, 1, N)
    rng = mwc.make_seeds(N, 99)
    dev_pts, dev_rng = pts.copy(), rng.copy()
    _lib.check(lib.fl_debug_apply_xf(mgr.fb.ctx, g, 5, 0, N, dev_pts.ctypes.data, dev_rng.ctypes.data))

    L = O.lib()
    L.ref_apply_xf.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    ref_pts, ref_rng = pts.copy(), rng.copy()
    P = params[5]
    # one oracle call for all points (a ctypes call per point was a tenth of the GPU suite's time)
    L.ref_apply_xf_n.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    assert L.ref_apply_xf_n(rdr.packer.prog.ctypes.data, P.ctypes.data, 0, N, ref_pts.ctypes.data, ref_rng.ctypes.data) == 0

    assert np.array_equal(dev_rng, ref_rng), 'RNG draws differ'
    assert np.array_equal(dev_pts[:, 2], ref_pts[:, 2]), 'colour blend differs'
    d, r = dev_pts[:, :2].astype(np.float64), ref_pts[:, :2].astype(np.float64)
    fin = np.isfinite(r).all(1) & (np.abs(r).max(1) < 1e6)
    both_bad = ~np.isfinite(d).all(1) & ~np.isfinite(r).all(1)
    ok = np.zeros(N, bool)
    ok[fin] = (np.abs(d[fin] - r[fin]) <= 2e-4 + 2e-3 * np.abs(r[fin])).all(1)
    ok |= both_bad | (~fin & ~both_bad & (np.abs(r).max(1) >= 1e6))
    def close(dv, rv):
        if not np.isfinite(rv).all() or np.abs(rv).max() >= 1e6:
            return not np.isfinite(dv).all() or np.abs(rv).max() >= 1e6
        return bool((np.abs(dv - rv) <= 2e-4 + 2e-3 * np.abs(rv)).all())

    def step(v, n):                                   # v moved by n float32 ulps
        v = np.float32(v)
        for _ in range(abs(n)):
            v = np.nextafter(v, np.float32(np.inf if n > 0 else -np.inf), dtype=np.float32)
        return float(v)

    bad = np.nonzero(~ok)[0]
    assert len(bad) <= 0.015 * N, (name, len(bad), d[bad][:3], r[bad][:3])
    unexplained = []
    for i in bad:
        # the oracle over 7 x 7 inputs within 6 ulp of the point: at a discontinuity (floor / trunc / compare) or where
        # the variation is ill-conditioned (popcorn's sin(tan(3y)) near a pole of tan: 1 ulp of y turns the sine by 0.1 rad)
        # its own results spread further than the tolerance; on a well-conditioned point they agree to a few ulp and a
        # wrong result fails
        outs = []
        for du in range(-6, 7, 2):
            for dv in range(-6, 7, 2):
                x, y, c = C.c_float(step(pts[i, 0], du)), C.c_float(step(pts[i, 1], dv)), C.c_float(pts[i, 2])
                st = rng[i:i + 1].copy()
                assert L.ref_apply_xf(rdr.packer.prog.ctypes.data, P.ctypes.data, 0, C.byref(x), C.byref(y), C.byref(c), st.ctypes.data) == 0
                outs.append((x.value, y.value))
        outs = np.array(outs, np.float64)
        if not np.isfinite(outs).all() or np.abs(outs).max() >= 1e6:
            continue                                   # the oracle itself leaves the finite range next to the point
        # well-conditioned point: the 49 results agree to a few ulp and the spread adds nothing to the tolerance; at a jump
        # the spread is the jump; where 1 ulp of input swings the result (sin(tan) next to a pole) the hardware's tan may
        # land anywhere within that swing of libm's
        # (x 4: the hardware's rcp / sqrt / sin of INTERMEDIATES are each a few ulp off libm's, which next to a pole — conic's
        # 1 / (1 + e cos t), condition number 1e5 — weighs like tens of ulp of the input)
        spread = outs.max(0) - outs.min(0)
        # A result magnified 1000-fold (conic's 1 / (1 + e cos t) where the sum cancels to 1e-5: the float32 rounding of
        # cos t alone is worth 0.6 % of it, whichever way the division that produced it was rounded) is held to 5 %.
        blown = np.abs(r[i]).max() >= 1e3 * max(1.0, float(np.abs(pts[i, :2]).max()))
        rel = 5e-2 if blown else 2e-3
        if not (np.isfinite(d[i]).all() and (np.abs(d[i] - r[i]) <= 2e-4 + rel * np.abs(r[i]) + 4.0 * spread).all()):
            unexplained.append(int(i))
    assert not unexplained, (name, 'points off the oracle at a well-conditioned input', unexplained[:5],
                             pts[unexplained[:3]], d[unexplained[:3]], r[unexplained[:3]])


def _ref_cases():
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'xf_apply.json')) as fp:
        g = json.load(fp)
    return g['npts'], g['cases']


@pytest.mark.parametrize('idx', range(99))
def test_xform_application_matches_reference_templates(mgr, idx):
    """The HIP xform application against vectors computed by the REFERENCE's own ``apply_xf`` template and variation entries
    (tests/golden/xf_apply.json, made by tests/golden/make_golden_xf.py: rendered and compiled as host C++), without the oracle in
    between: one xform per variation with every parameter off its default (every other one with a post affine), four xforms that
    sum three variations, 32 points each.  RNG streams bit-exact (same draws in the same order), colour to an ulp, points to the
    bar of the oracle comparison above (2e-3 relative / 2e-4 absolute: single hardware instructions for sin / cos / exp / log /
    rcp / sqrt) with at most 3 of 32 points of an ill-conditioned variation within 5 %."""
    import base64
    n, cases = _ref_cases()
    case = cases[idx]
    dec = lambda s, c: np.frombuffer(base64.b64decode(s), '<u4').reshape(n, c).copy()
    lib = _lib.load()
    gnm = {'type': 'animation', 'camera': {'scale': 0.25}, 'time': {'duration': 1, 'frame_width': 0.0},
           'palette': [[0.0] + configs.palette_encode(configs.grey_ramp())], 'xforms': {'0': dict(case['xform'], weight=1.0)}}
    prof = {'width': 64, 'height': 64, 'spp': 1, 'fps': 1, 'duration': 1, 'frame_width': 0}
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    g = rdr._handle(mgr.fb)
    mgr._copy(rdr, gnm)
    _lib.check(lib.fl_interp(mgr.fb.ctx, g, 64, 64, 0.5, 0.0))
    pts = np.zeros((n, 4), np.float32)
    pts[:, :3] = dec(case['points_in'], 3).view(np.float32)
    rng = dec(case['rng_in'], 3)
    _lib.check(lib.fl_debug_apply_xf(mgr.fb.ctx, g, 5, 0, n, pts.ctypes.data, rng.ctypes.data))
    want, want_rng = dec(case['points_out'], 3).view(np.float32), dec(case['rng_out'], 2)
    assert np.array_equal(rng[:, 1:], want_rng), 'random draws differ in number or order'
    assert np.abs(pts[:, 2] - want[:, 2]).max() <= 2.5e-7, 'colour blend'
    a, b = pts[:, :2].astype(np.float64), want[:, :2].astype(np.float64)
    fin = np.isfinite(b).all(1) & (np.abs(b).max(1) < 1e6)
    assert np.isfinite(a[fin]).all()
    err = np.abs(a[fin] - b[fin]) - (2e-4 + 2e-3 * np.abs(b[fin]))
    loose = np.abs(a[fin] - b[fin]) - (2e-4 + 5e-2 * np.abs(b[fin]))
    nbad = int((err > 0).any(1).sum())
    assert nbad <= 3 and not (loose > 0).any(), ('+'.join(case['variations']), nbad, a[fin][(err > 0).any(1)][:3], b[fin][(err > 0).any(1)][:3])
