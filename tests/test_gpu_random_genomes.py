    # (seed 226 — handkerchief + perspective + rings2, from the soak of seeds 221-260: block L1 0.20 against 8
    # trajectories, 0.049 against 64, 0.033 against 512, 0.0235 against 4096 and below the bar against 65536 trajectories
    # of 1024 iterations)
"""
Randomised genomes through the C ABI: the structure of the genome is data here (one precompiled
interpreter kernel), so parity must hold for ANY structure, not just the BASELINE configs.
A seeded generator draws xform counts, variation sets (with parameters), post affines, final
xforms, animated ([p0, v0, p1, v1] + extra knots) splines and two palettes; for each genome:
  * the interpolated parameter blocks (one per walker slot) agree with the oracle's by-name float64 restatement;
  * every word of the block that is not a spline / precalc value is exactly the structure the
    program says (variation numbers, counts, zero padding);
  * a short iterate lands the same fraction of samples in frame as the oracle's flam3-style game
    driven by the ORACLE's parameter blocks, with matching mean colour (distributional, since
    random variations use hardware transcendentals).
48 seeds are committed; seeds 17-336 were run once by hand (tools/diag_random_seed.py, DESIGN.md §5 'Fuse'):
that run found the point-attractor overflow of the tile accumulate (fixed; test_gpu_parity.py
test_binned_point_attractor_keeps_every_sample) and six single-xform genomes that are not a fair
comparison (a lone map does not mix: eight CPU trajectories are not a distribution).
"""
import ctypes as C

import numpy as np
import pytest

from common import O, prepare, frame_times
from cuburn_amd import configs, profile, render, _lib
from cuburn_amd.genome import variations as V

pytestmark = pytest.mark.gpu
NSLOTS = 1024

# variations that are safe to draw at random weights (bounded output for bounded input or
# at least not systematically divergent); the full set is covered one by one in
# test_gpu_variations.py
POOL = ['linear', 'sinusoidal', 'spherical', 'swirl', 'horseshoe', 'polar', 'handkerchief', 'heart', 'disc',
        'spiral', 'hyperbolic', 'diamond', 'ex', 'julia', 'bent', 'waves', 'fisheye', 'popcorn', 'power',
        'cosine', 'rings', 'fan', 'blob', 'pdj', 'fan2', 'rings2', 'eyefish', 'bubble', 'cylinder',
        'perspective', 'julian', 'juliascope', 'ngon', 'curl', 'rectangles', 'tangent', 'cross', 'cell',
        'bipolar', 'wedge', 'scry', 'split', 'stripes', 'waves2', 'mobius', 'flux']


@pytest.fixture(scope='module')
def mgr():
    from __graft_entry__ import build
    build()
    return render.RenderManager(device=0, nslots=NSLOTS, host_seed=23)


@pytest.fixture(scope='module')
def mgr_prod(mgr):
    """Production geometry (RenderManager() defaults: 1536 slots): used for the animated genomes."""
    return render.RenderManager(device=0, host_seed=23)


def random_spline(rs, lo, hi, animated):
    a = float(rs.uniform(lo, hi))
    if not animated or rs.rand() < 0.5:
        return a
    b = float(rs.uniform(lo, hi))
    span = hi - lo
    knots = [a, float(rs.uniform(-0.3, 0.3) * span), b, float(rs.uniform(-0.3, 0.3) * span)]
    if rs.rand() < 0.4:                      # an interior knot
        knots += [float(rs.uniform(0.2, 0.8)), float(rs.uniform(lo, hi))]
    return knots


def random_genome(seed):
    rs = np.random.RandomState(seed)
    animated = seed % 2 == 1
    nxf = int(rs.randint(1, 13))
    xforms = {}
    for i in range(nxf):
        names = list(rs.choice([n for n in POOL if n in V.var_ids], size=int(rs.randint(1, 4)), replace=False))
        if rs.rand() < 0.5 and 'linear' not in names:
            names.append('linear')
        var = {}
        for n in names:
            var[n] = {'weight': random_spline(rs, 0.2, 0.9, animated)}
            for p, (dflt, interp) in V.var_params[n].items():
                if p != 'weight' and rs.rand() < 0.7:
                    base = dflt if dflt else 0.5
                    var[n][p] = random_spline(rs, 0.5 * base, 1.5 * base, animated) if interp == 'mag' else \
                        random_spline(rs, -abs(base), abs(base), animated)
        xf = {'weight': random_spline(rs, 0.1, 1.0, animated), 'color': random_spline(rs, 0.0, 1.0, animated),
              'color_speed': random_spline(rs, 0.1, 0.9, animated),
              'pre_affine': configs._affine(float(rs.uniform(-180, 180)), float(rs.uniform(0.3, 0.8)),
                                            float(rs.uniform(-0.6, 0.6)), float(rs.uniform(-0.5, 0.5))),
              'variations': var}
        if animated and rs.rand() < 0.5:
            xf['pre_affine']['angle'] = [xf['pre_affine']['angle'], float(rs.uniform(-90, 90)),
                                         xf['pre_affine']['angle'] + float(rs.uniform(-60, 60)), float(rs.uniform(-90, 90))]
        if rs.rand() < 0.4:
            xf['post_affine'] = configs._affine(float(rs.uniform(-30, 30)), float(rs.uniform(0.8, 1.1)),
                                                float(rs.uniform(-0.1, 0.1)), float(rs.uniform(-0.1, 0.1)))
        xforms[str(i)] = xf
    gnm = {'type': 'animation', 'name': 'random-%d' % seed,
           'camera': {'center': {'x': random_spline(rs, -0.2, 0.2, animated), 'y': random_spline(rs, -0.2, 0.2, animated)},
                      'rotation': random_spline(rs, -40, 40, animated), 'scale': random_spline(rs, 0.15, 0.35, animated)},
           'time': {'duration': 2, 'frame_width': 1.0 if animated else 0.0},
           'palette': [[0.0] + configs.palette_encode(configs.grey_ramp())],
           'xforms': xforms}
    if animated:
        ramp = configs.grey_ramp()[::-1].copy()
        gnm['palette'].append([1.0] + configs.palette_encode(ramp))
    if rs.rand() < 0.4:
        gnm['final_xform'] = {'color': 0.2, 'color_speed': float(rs.uniform(0, 0.5)),
                              'pre_affine': configs._affine(float(rs.uniform(-10, 10)), float(rs.uniform(0.9, 1.1)), 0.0, 0.0),
                              'variations': {'linear': {'weight': 1.0}}}
    prof = {'width': 384, 'height': 216, 'spp': 2 ** 24 / (384.0 * 216.0), 'fps': 24, 'duration': 2,
            'frame_width': 1.0 if animated else 0.0, 'output': {'type': 'raw'},
            'filter_order': ['bilateral', 'logscale', 'colorclip']}
    return gnm, prof


def _seeds():
    """1 .. 48 by default; FLAME_SOAK_SEEDS=a-b widens the range for a soak run (profiles/r06_soak_random_genomes.txt: 49-248)."""
    import os
    e = os.environ.get('FLAME_SOAK_SEEDS')
    if e:
        a, b = e.split('-')
        return list(range(int(a), int(b) + 1))
    return list(range(1, 49))


@pytest.mark.parametrize('seed', _seeds())
def test_random_genome_parity(mgr, mgr_prod, seed):
    lib = _lib.load()
    gnm, prof = random_genome(seed)
    if seed % 2 == 1:                       # animated genomes run in the geometry that ships
        mgr = mgr_prod
        assert mgr.fb.nslots == 1536
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    g = rdr._handle(mgr.fb)
    mgr._copy(rdr, gnm)
    dim = mgr.fb.calc_dim(gprof.width, gprof.height)
    tc = 0.37
    ts, td = frame_times(gprof, tc)
    _lib.check(lib.fl_interp(mgr.fb.ctx, g, dim.w, dim.h, ts, td))
    dev = mgr.fb.read('params', (mgr.fb.nslots, rdr.packer.pstride), np.float32, g)
    F = prepare(gnm, prof, tc, nslots=mgr.fb.nslots)
    ref = F['params']
    names = ['.'.join(n) for n in rdr.packer.packed]
    lastden = names.index('den.' + rdr.packer.xform_keys[-1])
    assert np.all(dev[:, lastden] >= 1.0)
    dev[:, lastden] = ref[:, lastden]
    structural = np.array([n.split('.')[-1].startswith('#') or n.startswith('pad') for n in names])
    assert np.array_equal(dev[:, structural].view(np.uint32), ref[:, structural].view(np.uint32)), 'structure words'
    err = np.abs(dev - ref)[:, ~structural] / (np.abs(ref[:, ~structural]) + 1.0)
    i = np.unravel_index(np.argmax(err), err.shape)
    assert err.max() < 5e-5, (np.array(names)[~structural][i[1]], err.max())

    # short iterate vs the flam3-style game on the oracle's blocks
    n = 2 ** 24
    run = C.c_uint64()
    _lib.check(lib.fl_iterate(mgr.fb.ctx, g, dim.w, dim.h, float(n), mgr.fuse, 1, C.byref(run)))     # the default schedule (fuse 256)
    nbins = dim.ah * dim.astride
    front = mgr.fb.read('front', (nbins, 4), np.float32).astype(np.float64)
    assert np.isfinite(front).all()
    refh, _, _ = O.flam3_render(F['dim'], F['packer'].prog, F['params'], F['palette'], F['seeds'], n, 8)
    refh = refh.astype(np.float64)
    fg, fr = front[:, 3].sum() / run.value, refh[:, 3].sum() / n
    assert abs(fg - fr) < 0.01 + 0.02 * fr, (fg, fr)
    if fr > 0.02:
        cg, cr = front[:, :3].sum(0) / front[:, 3].sum(), refh[:, :3].sum(0) / refh[:, 3].sum()
        assert np.abs(cg - cr).max() < 2.5 / 255, (cg, cr)
        H, W = dim.ah // 16 * 16, dim.astride // 16 * 16
        def blocks(a):
            return a[:, 3].reshape(dim.ah, dim.astride)[:H, :W].reshape(H // 16, 16, W // 16, 16).sum((1, 3))
        bg, br = blocks(front), blocks(refh)
        l1 = np.abs(bg / bg.sum() - br / br.sum()).sum()
        # the bar is the shot-noise floor of the two samples, sum over blocks of sqrt(2 / pi * p * (1/Ng + 1/Nr)), times 3
        # (wave-coherent xform choice and the CPU game's eight long trajectories are both clustered samples: block z-scores
        # of 1.0-1.2 against an independent game) plus 2 % — as tests/test_gpu_fullsize.py; round 3: 5 % flat
        p = br / br.sum()
        floor = np.sqrt(2 / np.pi * p * (1.0 / bg.sum() + 1.0 / br.sum())).sum()
        assert l1 < 0.02 + 3 * floor, (l1, floor)


def _blocks16(a, dim):
    H, W = dim.ah // 16 * 16, dim.astride // 16 * 16
    return a.reshape(dim.ah, dim.astride)[:H, :W].reshape(H // 16, 16, W // 16, 16).sum((1, 3))


# Genomes of ONE xform: a lone map converges to its orbit at its own contraction rate, with no averaging
# over xform choices — the class for which a fuse of 64 was measurably too short (seed 209: cross + fan2 +
# linear, block L1 0.11 at fuse 64, 0.010 at 256).  Everything here runs at RenderManager's DEFAULTS,
# through queue_frame: no explicit fuse, production slots (1024 of them for these sample counts).  The CPU sample
# uses 4096 trajectories (eight trajectories of a map that does not mix are not a distribution, and 64 are a poor
# one: seed 176, cell + julian, is 0.046 from a 64-trajectory sample and 0.021 from 4096 — the error of the CPU
# sample's basin weights); the two seeds that mix least get 65536 short trajectories, a sample made like the GPU's.
@pytest.mark.parametrize('seed', [209, 41, 48, 50, 102, 110, 176, 226, 270, 278, 295])
def test_single_xform_genome_default_schedule(seed):
    gnm, prof = random_genome(seed)
    assert len(gnm['xforms']) == 1
    prof = dict(prof, filter_order=[])                       # the accumulator itself is compared
    gprof = profile.wrap(prof, gnm)
    m = render.RenderManager(device=0, host_seed=23)
    assert m.fuse == 256 and m.fb.nslots == 1536             # cuburn/render.py:215 (1536 until the first frame's sample count is known)
    rdr = render.Renderer(gnm, gprof)
    tc = 0.37
    evt, _ = m.queue_frame(rdr, gnm, gprof, tc)
    evt.synchronize()
    dim = m.fb.calc_dim(gprof.width, gprof.height)
    dens = m.fb.read('front', (dim.ah * dim.astride, 4), np.float32).astype(np.float64)[:, 3]
    nrun = m.last_nsamples
    F = prepare(gnm, prof, tc, nslots=m.fb.nslots)
    n = 2 ** 26
    # (seed 226 — handkerchief + perspective + rings2, from the soak of seeds 221-260 — mixes even less: block L1 0.20
    # against 8 trajectories, 0.049 against 64, 0.033 against 512, 0.0235 against 4096 and below the bar against 65536
    # trajectories of 1024 iterations, i.e. against a CPU sample made the way the GPU's is: many short orbits)
    refh, _, _ = O.flam3_render(F['dim'], F['packer'].prog, F['params'], F['palette'], F['seeds'], n, 65536 if seed in (226, 176) else 4096)
    refd = refh.astype(np.float64)[:, 3]
    fg, fr = dens.sum() / nrun, refd.sum() / n
    assert abs(fg - fr) < 0.005 + 0.01 * fr, (fg, fr)
    if fr > 0.02:
        bg, br = _blocks16(dens, dim), _blocks16(refd, dim)
        l1 = np.abs(bg / bg.sum() - br / br.sum()).sum()
        # shot noise of both samples: sum over blocks of sqrt(2 / pi * p * (1/Ng + 1/Nr))
        p = br / br.sum()
        floor = np.sqrt(2 / np.pi * p * (1.0 / dens.sum() + 1.0 / refd.sum())).sum()
        assert l1 < 0.02 + 3 * floor, (l1, floor)
    m.fb.free()


def _variation_genome(names):
    """One xform per variation (plus a little linear so that points keep moving), default parameters."""
    xforms = {}
    for i, n in enumerate(names):
        var = {n: dict([(k, dv if dv else 0.6) for k, (dv, _) in V.var_params[n].items()], weight=0.6)}
        if n != 'linear':
            var['linear'] = {'weight': 0.4}
        xforms['%02d' % i] = {'weight': 1.0, 'color': i / max(len(names) - 1, 1), 'color_speed': 0.5,
                              'pre_affine': configs._affine(17.0 * i, 0.7, 0.3 * np.cos(i), 0.3 * np.sin(i)),
                              'variations': var}
    gnm = {'type': 'animation', 'name': 'vars', 'camera': {'center': {'x': 0.0, 'y': 0.0}, 'rotation': 0.0, 'scale': 0.25},
           'time': {'duration': 1, 'frame_width': 0.0}, 'palette': [[0.0] + configs.palette_encode(configs.grey_ramp())],
           'xforms': xforms}
    prof = {'width': 384, 'height': 216, 'spp': 100.0, 'fps': 1, 'duration': 1, 'frame_width': 0.0,
            'output': {'type': 'raw'}, 'filter_order': ['bilateral', 'logscale', 'colorclip']}
    return gnm, prof


ALL_VARS = sorted(V.var_ids, key=lambda n: V.var_ids[n])


@pytest.mark.parametrize('chunk', list(range(0, len(ALL_VARS), 8)))
def test_every_variation_per_genome_kernel_equals_interpreter(chunk, monkeypatch):
    """All 95 variations, eight per genome: the kernel compiled for the genome's structure (hipRTC)
    and the precompiled interpreter kernel (FLAME_RTC=0) leave identical RNG states, walker points,
    sample counters and per-cell densities — the specialisation inlines every variation body with a constant
    id where the interpreter calls one out-of-line function, and must not change a single bit."""
    lib = _lib.load()
    names = ALL_VARS[chunk:chunk + 8]
    gnm, prof = _variation_genome(names)
    gprof = profile.wrap(prof, gnm)
    out = {}
    for rtc in ('0', '1'):
        monkeypatch.setenv('FLAME_RTC', rtc)
        m = render.RenderManager(device=0, nslots=NSLOTS, host_seed=5)
        rdr = render.Renderer(gnm, gprof)
        g = rdr._handle(m.fb)
        m._copy(rdr, gnm)
        dim = m.fb.calc_dim(gprof.width, gprof.height)
        _lib.check(lib.fl_interp(m.fb.ctx, g, dim.w, dim.h, 0.5, 0.0))
        _lib.check(lib.fl_debug_clear(m.fb.ctx, dim.w, dim.h, 1))
        _lib.check(lib.fl_debug_iter_launch(m.fb.ctx, g, dim.w, dim.h, 0, 20, 4, 1))
        ctr = np.zeros(4, np.uint64)
        _lib.check(lib.fl_debug_counters(m.fb.ctx, ctr.ctypes.data))
        _lib.check(lib.fl_debug_flush(m.fb.ctx, dim.w, dim.h))
        out[rtc] = (ctr, m.fb.read('seeds', (m.fb.nwalkers, 3), np.uint32),
                    m.fb.read('points', (NSLOTS * 256, 4), np.float32),
                    m.fb.read('front', (dim.ah * dim.astride, 4), np.float32))
        m.fb.free()
    (ca, ra, pa, fa), (cb, rb, pb, fb) = out['0'], out['1']
    assert np.array_equal(ca, cb), (names, ca, cb)
    assert np.array_equal(ra, rb), names
    both_nan = np.isnan(pa) & np.isnan(pb)
    assert np.array_equal(pa.view(np.uint32)[~both_nan], pb.view(np.uint32)[~both_nan]), names
    # densities are integers (exact); colour sums are float atomics whose order varies run to run
    assert np.array_equal(fa[:, 3], fb[:, 3]), names
    assert np.allclose(fa[:, :3], fb[:, :3], rtol=1e-5, atol=1e-5), names
