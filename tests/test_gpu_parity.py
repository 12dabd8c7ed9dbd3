"""
GPU parity tests: the HIP path (through the C ABI of libflame_hip.so) against the CPU oracle.

Bit-exact: RNG-driven integer work (shuffle permutation, packed palette, packed histogram of a
flame without transcendentals, flush + hot flags, output dither).  Tolerance (stated per test):
float parameter preparation and every filter.  Distributional: histograms of flames whose
variations use hardware transcendentals.
"""
import ctypes as C
import os

import numpy as np
import pytest

from common import O, prepare, frame_times
from cuburn_amd import configs, profile, render, _lib
from cuburn_amd.packer import GenomePacker

pytestmark = pytest.mark.gpu

NSLOTS = 1024


@pytest.fixture(scope='module')
def mgr(built):
    return render.RenderManager(device=0, nslots=NSLOTS, host_seed=42)


@pytest.fixture(scope='module')
def mgr_prod(built):
    """The geometry that ships: RenderManager() defaults (1536 slots of 4 waves up to ~1440p)."""
    m = render.RenderManager(device=0, host_seed=42)
    assert (m.fb.nw, m.fb.nslots) == render.Framebuffers.NARROW == (4, 1536)
    return m


def walkers(mgr):
    """(slots, threads per slot, walker count, oracle geometry) of a manager's current context."""
    ns, nt = mgr.fb.nslots, mgr.fb.nthreads
    return ns, nt, ns * nt, {256: O.GEOM_4x64, 512: O.GEOM_8x64, 1024: O.GEOM_16x64}[nt]


def small(cfg, w, h, **kw):
    gnm, prof = cfg(**kw)
    prof = dict(prof, width=w, height=h)
    return gnm, prof


def setup_frame(mgr, gnm, prof, tc=0.5):
    """Upload + interp on the GPU; returns (rdr, gprof, dim, handle) and device params/palette."""
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    g = rdr._handle(mgr.fb)
    mgr._copy(rdr, gnm)
    dim = mgr.fb.calc_dim(gprof.width, gprof.height)
    ts, td = frame_times(gprof, tc)
    _lib.check(_lib.load().fl_interp(mgr.fb.ctx, g, dim.w, dim.h, ts, td))
    return rdr, gprof, dim, g, ts, td


def test_shuffle_bit_exact(mgr):
    lib = _lib.load()
    for rnd in range(7):
        out = np.zeros(256, np.uint32)
        _lib.check(lib.fl_debug_shuffle(mgr.fb.ctx, rnd, out.ctypes.data))
        assert np.array_equal(out, O.shuffle_perm(O.GEOM_4x64, rnd)), rnd
        assert np.array_equal(np.sort(out), np.arange(256))


@pytest.mark.parametrize('cfg,prod', [('cfg2', False), ('cfg3', False), ('cfg5', False), ('cfg3', True)])
def test_interp_params(mgr, mgr_prod, cfg, prod):
    """One parameter block per walker slot (block s at ts + s*td/nslots), for 1024 slots and for
    the production 1536."""
    mgr = mgr_prod if prod else mgr
    gnm, prof = small(configs.CONFIGS[cfg], 640, 360)
    rdr, gprof, dim, g, ts, td = setup_frame(mgr, gnm, prof, 0.3)
    F = prepare(gnm, prof, 0.3, nslots=mgr.fb.nslots)
    dev = mgr.fb.read('params', (mgr.fb.nslots, rdr.packer.pstride), np.float32, g)
    ref = F['params']
    names = ['.'.join(n) for n in rdr.packer.packed]
    lastden = names.index('den.' + rdr.packer.xform_keys[-1])
    assert np.all(dev[:, lastden] >= 1.0)
    dev[:, lastden] = ref[:, lastden]
    # float32 device evaluation vs float64 formulas: 1e-5 relative (+1e-5 absolute for the
    # pixel-scale camera offsets which are O(1e3))
    err = np.abs(dev - ref) / (np.abs(ref) + 1.0)
    i = np.unravel_index(np.argmax(err), err.shape)
    assert err.max() < 2e-5, (names[i[1]], dev[i], ref[i])


def test_palette_bit_exact(mgr):
    gnm, prof = small(configs.cfg3, 640, 360)
    # the palette kernel advances its RNG states: take them from the device first
    ns, nt, nwalk, _ = walkers(mgr)
    seeds = mgr.fb.read('seeds', (nwalk + 64 * 256, 3), np.uint32)
    rdr, gprof, dim, g, ts, td = setup_frame(mgr, gnm, prof, 0.4)
    dev = mgr.fb.read('palette', (64, 256), np.uint64)
    from common import oracle_palette
    ref, rng_after = oracle_palette(gnm, np.float32(ts), np.float32(td), seeds[nwalk:])
    assert np.array_equal(dev, ref)
    seeds2 = mgr.fb.read('seeds', (nwalk + 64 * 256, 3), np.uint32)
    assert np.array_equal(seeds2[nwalk:], rng_after)


def run_device_model(mgr, gnm, prof, nrounds, fuse, launches=1, mode=0, seeds_in=None, tc=0.5):
    """GPU and oracle from the same device params / palette / seeds / points; returns both states.
    mode 0 = packed global atomics (hot flags + roulette live), 1 = binned (no sample thinning)."""
    lib = _lib.load()
    ns, nt, nwalk, geom = walkers(mgr)
    if seeds_in is not None:                # before interp: the palette kernel draws from these too
        mgr.fb.write('seeds', seeds_in)
    seeds0 = mgr.fb.read('seeds', (nwalk + 64 * 256, 3), np.uint32)
    rdr, gprof, dim, g, ts, td = setup_frame(mgr, gnm, prof, tc)
    d = O.calc_dim(dim.w, dim.h)
    nbins = dim.ah * dim.astride
    _lib.check(lib.fl_debug_clear(mgr.fb.ctx, dim.w, dim.h, 1))
    params = mgr.fb.read('params', (ns, rdr.packer.pstride), np.float32, g)
    palette = mgr.fb.read('palette', (64, 256), np.uint64)
    seeds = mgr.fb.read('seeds', (nwalk + 64 * 256, 3), np.uint32)
    rng = seeds[:nwalk].copy()
    points = np.full((nwalk, 4), np.nan, np.float32)
    hot = np.zeros(nbins // 16, np.uint32)
    atom = np.zeros(nbins, np.uint64)
    out4 = np.zeros((nbins, 4), np.float32)
    res = []
    r0 = 0
    for k in range(launches):
        f = fuse if k == 0 else 0
        _lib.check(lib.fl_debug_iter_launch(mgr.fb.ctx, g, dim.w, dim.h, r0, nrounds + f, f, mode))
        ctr_ref = O.iter_launch(geom, d, rdr.packer.prog, params, palette, rng, points, ns,
                                hot, atom, out4, r0, nrounds + f, f)
        ctr_dev = np.zeros(4, np.uint64)
        _lib.check(lib.fl_debug_counters(mgr.fb.ctx, ctr_dev.ctypes.data))
        dev_atom = mgr.fb.read('atom', (nbins,), np.uint64)
        res.append(dict(ctr_ref=ctr_ref.copy(), ctr_dev=ctr_dev, atom_ref=atom.copy(), atom_dev=dev_atom))
        if mode == 1:                       # binned mode never thins: flags are neither read nor kept
            _lib.check(lib.fl_debug_clear_hot(mgr.fb.ctx, dim.w, dim.h))
            hot[:] = 0
        _lib.check(lib.fl_debug_flush(mgr.fb.ctx, dim.w, dim.h))
        O.flush(d, atom, out4, hot)
        if mode == 1:
            _lib.check(lib.fl_debug_clear_hot(mgr.fb.ctx, dim.w, dim.h))
            hot[:] = 0
        res[-1].update(front_dev=mgr.fb.read('front', (nbins, 4), np.float32), front_ref=out4.copy(),
                       hot_dev=mgr.fb.read('hot', (nbins // 16,), np.uint32), hot_ref=hot.copy())
        r0 += nrounds + f
    dev_rng = mgr.fb.read('seeds', (nwalk + 64 * 256, 3), np.uint32)[:nwalk]
    dev_pts = mgr.fb.read('points', (nwalk, 4), np.float32)
    return res, (rng, points), (dev_rng, dev_pts), dim, seeds0


def linear_flame():
    """3 linear xforms + post affine + final xform: no transcendentals anywhere => bit-exact."""
    gnm, prof = configs.cfg2()
    for k in gnm['xforms']:
        gnm['xforms'][k]['variations'] = {'linear': {'weight': 0.8}, 'bent': {'weight': 0.2}}
    gnm['xforms']['1']['post_affine'] = configs._affine(10.0, 0.9, 0.05, -0.1)
    gnm['final_xform'] = {'color': 0.3, 'color_speed': 0.25, 'pre_affine': configs._affine(-8.0, 0.97, 0.02, 0.01),
                          'variations': {'linear': {'weight': 1.0}}}
    return gnm, dict(prof, width=1280, height=720)


def test_iter_bit_exact_linear(mgr):
    gnm, prof = linear_flame()
    res, ref_state, dev_state, dim, _ = run_device_model(mgr, gnm, prof, nrounds=8, fuse=5, launches=2)
    for k, r in enumerate(res):
        assert int(r['ctr_dev'][3]) == 0 and int(r['ctr_ref'][3]) == 0, 'spill path taken; shrink the test'
        assert np.array_equal(r['ctr_dev'][:3], r['ctr_ref'][:3]), (k, r['ctr_dev'], r['ctr_ref'])
        assert np.array_equal(r['atom_dev'], r['atom_ref']), k
        # conservation: sum of counts == accepted samples
        assert int((r['atom_dev'] >> np.uint64(54)).sum()) == int(r['ctr_dev'][0])
        assert np.array_equal(r['front_dev'], r['front_ref']), k
        assert np.array_equal(r['hot_dev'], r['hot_ref']), k
    assert np.array_equal(dev_state[0], ref_state[0])          # RNG states after both launches
    assert np.array_equal(dev_state[1][:, :3], ref_state[1][:, :3])


def animated_linear_flame():
    """The transcendental-free flame in motion: camera pan, one rotating xform, two palettes, a
    frame window that spans the whole animation: every slot sees a different parameter block
    and the 64 palette rows differ."""
    gnm, prof = linear_flame()
    gnm['camera']['center'] = {'x': [-0.15, 0.3, 0.15, 0.3], 'y': 0.0}
    gnm['camera']['scale'] = 1.0          # zoomed in: every cell stays below the packed-add limit of the binned drain
    gnm['xforms']['0']['pre_affine']['angle'] = [70.0, 30.0, 100.0, 30.0]
    gnm['palette'] = [configs._pal(0.0, configs.fire_palette()), configs._pal(1.0, configs.ice_palette())]
    gnm['time'] = {'duration': 1, 'frame_width': 1.0}
    return gnm, dict(prof, frame_width=1.0, fps=1, duration=1)


@pytest.mark.parametrize('mode', [0, 1])
def test_iter_bit_exact_animated_production_slots(mgr_prod, mode):
    """Temporal sampling in the geometry that ships (1536 slots): slot s iterates parameter block s
    and palette row s*64/1536; packed histogram, counters, RNG and walkers equal the oracle's device
    model, with direct atomics and through the binned accumulate (whose drain kernel derives the
    palette row of every batch on its own)."""
    gnm, prof = animated_linear_flame()
    res, ref_state, dev_state, dim, _ = run_device_model(mgr_prod, gnm, prof, nrounds=19, fuse=5, launches=2, mode=mode)
    for k, r in enumerate(res):
        assert int(r['ctr_dev'][3]) == 0 and int(r['ctr_ref'][3]) == 0, 'spill path taken; shrink the test'
        assert np.array_equal(r['ctr_dev'][:3], r['ctr_ref'][:3]), (k, r['ctr_dev'], r['ctr_ref'])
        assert np.array_equal(r['atom_dev'], r['atom_ref']), k
        assert np.array_equal(r['front_dev'], r['front_ref']), k
    assert np.array_equal(dev_state[0], ref_state[0])
    assert np.array_equal(dev_state[1][:, :3], ref_state[1][:, :3])


def test_temporal_samples_equally_weighted(built):
    """
    Every temporal sample of the frame window must receive the same number of iterations (the
    reference runs one block column per temporal sample: cuburn/render.py:343-346,
    cuburn/code/iter.py:165,184).  A flame whose only motion is a linear camera pan, rendered by a
    default RenderManager() at 1920x1080 with the window spanning the whole pan: the density
    centroid along the pan must sit where the still frame at the window centre has it, and the
    pan must add the variance of a UNIFORM distribution over the window (T^2/12).  (A 2:1
    weighting of the window halves, which a 1536-slot geometry gives when slots are mapped onto
    1024 samples modulo 1024, moves the centroid by T/12 = 5.8 px here and is caught.)
    """
    lib = _lib.load()
    gnm, prof = configs.cfg2(samples=2 ** 26)
    gnm['camera'] = {'center': {'x': [-0.15, 0.3, 0.15, 0.3], 'y': 0.0}, 'rotation': 0.0, 'scale': 0.12}
    gnm['time'] = {'duration': 1, 'frame_width': 1.0}
    still = dict(prof, frame_width=0.0, fps=1, duration=1)
    moving = dict(prof, frame_width=1.0, fps=1, duration=1)
    m = render.RenderManager(device=0, host_seed=42)                 # production defaults
    assert m.fb.nslots == 1536
    pan_px = 0.3 * 0.12 * 1920

    def moments(prof_):
        gprof = profile.wrap(prof_, gnm)
        rdr = render.Renderer(gnm, gprof)
        dim = m.fb.calc_dim(gprof.width, gprof.height)
        ts, td = frame_times(gprof, 0.5)
        g = rdr._handle(m.fb)
        m._copy(rdr, gnm)
        _lib.check(lib.fl_interp(m.fb.ctx, g, dim.w, dim.h, ts, td))
        run = C.c_uint64()
        _lib.check(lib.fl_iterate(m.fb.ctx, g, dim.w, dim.h, float(2 ** 26), 64, m.resolve_accum_mode(dim), C.byref(run)))
        front = m.fb.read('front', (dim.ah * dim.astride, 4), np.float32)
        d = density(front, dim)
        # central window only: the flame's far tails leave the frame on one side first
        col = d.sum(0)
        x = np.arange(dim.astride, dtype=np.float64)
        mean = (col * x).sum() / col.sum()
        var = (col * (x - mean) ** 2).sum() / col.sum()
        return mean, var, td, d.sum() / run.value

    mean_s, var_s, td_s, in_s = moments(still)
    mean_m, var_m, td_m, in_m = moments(moving)
    assert td_s == 0 and td_m == 1.0
    print('centroid still %.3f moving %.3f  variance added %.1f (uniform window: %.1f)  in-frame %.4f %.4f' % (
        mean_s, mean_m, var_m - var_s, pan_px ** 2 / 12, in_s, in_m))
    assert in_s > 0.94 and in_m > 0.94, (in_s, in_m)                 # only far tails are clipped
    assert abs(mean_m - mean_s) < 0.75, (mean_m, mean_s, pan_px / 12)
    added = var_m - var_s
    assert abs(added - pan_px ** 2 / 12) < 0.06 * pan_px ** 2 / 12 + 3.0, (added, pan_px ** 2 / 12)
    m.fb.free()


@pytest.mark.parametrize('which', ['linear', 'cfg3', 'cfg5'])
def test_per_genome_kernel_equals_interpreter(built, which, monkeypatch, capfd):
    """fl_iterate runs a kernel compiled for the genome's structure (hipRTC, the counterpart of the
    reference's per-genome CUDA module, render.py:232-236); FLAME_RTC=0 runs the precompiled
    interpreter kernel.  Same arithmetic in the same order: packed histograms, sample counters,
    RNG states and walker points must agree bit for bit — with hardware transcendentals too —
    in both accumulate modes."""
    if which == 'linear':
        gnm, prof = animated_linear_flame()
    else:
        gnm, prof = small(configs.CONFIGS[which], 640, 360)
        gnm['camera']['scale'] = 1.0 if which == 'cfg3' else 0.8
    out = {}
    for rtc in ('0', '1'):
        monkeypatch.setenv('FLAME_RTC', rtc)
        m = render.RenderManager(device=0, host_seed=42)            # production geometry
        for mode in (0, 1):
            lib = _lib.load()
            rdr, gprof, dim, g, ts, td = setup_frame(m, gnm, prof, 0.4)
            nbins = dim.ah * dim.astride
            _lib.check(lib.fl_debug_clear(m.fb.ctx, dim.w, dim.h, 1))
            _lib.check(lib.fl_debug_iter_launch(m.fb.ctx, g, dim.w, dim.h, 0, 24, 5, mode))
            ctr = np.zeros(4, np.uint64)
            _lib.check(lib.fl_debug_counters(m.fb.ctx, ctr.ctypes.data))
            _lib.check(lib.fl_debug_flush(m.fb.ctx, dim.w, dim.h))
            out[rtc, mode] = dict(ctr=ctr, front=m.fb.read('front', (nbins, 4), np.float32),
                                  rng=m.fb.read('seeds', (m.fb.nwalkers, 3), np.uint32),
                                  pts=m.fb.read('points', (m.fb.nslots * m.fb.nthreads, 4), np.float32))
        m.fb.free()
    assert 'interpreter kernel' not in capfd.readouterr().err, 'the per-genome kernel was not used'
    for mode in (0, 1):
        a, b = out['0', mode], out['1', mode]
        assert int(a['ctr'][0]) > 100000
        assert np.array_equal(a['ctr'], b['ctr']), (which, mode, a['ctr'], b['ctr'])
        assert np.array_equal(a['rng'], b['rng']), (which, mode)
        assert np.array_equal(a['pts'].view(np.uint32), b['pts'].view(np.uint32)), (which, mode)
        assert np.array_equal(a['front'][:, 3], b['front'][:, 3]), (which, mode)
        if mode == 1 or int(a['ctr'][3]) == 0:          # (drains of full cells regroup float adds in atomic mode)
            assert np.array_equal(a['front'].view(np.uint32), b['front'].view(np.uint32)), (which, mode)


def test_pipelined_launches_equal_serial_launches(built, monkeypatch):
    """A frame of several launches runs the tile accumulate + flush of launch k on the lane's aux
    stream while launch k+1 iterates (two log / directory sets; the reference alternates two streams
    the same way, render.py:340-369).  FLAME_NO_INTRA_OVERLAP=1 keeps everything on one stream.
    Same kernels on the same data in the same order: the density channel must be identical, colour
    sums equal up to the order of float adds of drained cells."""
    gnm, prof = small(configs.cfg3, 640, 360)
    lib = _lib.load()
    outs = {}
    for serial in ('0', '1'):
        monkeypatch.setenv('FLAME_NO_INTRA_OVERLAP', serial)
        m = render.RenderManager(device=0, host_seed=42)
        rdr, gprof, dim, g, ts, td = setup_frame(m, gnm, prof, 0.4)
        run = C.c_uint64()
        nsamp = float(5 * 1024 * m.fb.nslots * m.fb.nthreads)          # 5120 rounds: the reference's schedule saves a launch (1024 + 1536 + 2304 + 256)
        _lib.check(lib.fl_iterate(m.fb.ctx, g, dim.w, dim.h, nsamp, 64, 1, C.byref(run)))
        outs[serial] = (run.value, m.fb.read('front', (dim.ah * dim.astride, 4), np.float32),
                        m.fb.read('seeds', (m.fb.nwalkers, 3), np.uint32))
        t = m.timings()
        assert t['launches'] == 4
        m.fb.free()
    (na, fa, ra), (nb, fb, rb) = outs['0'], outs['1']
    assert na == nb and np.array_equal(ra, rb)
    assert np.array_equal(fa[:, 3], fb[:, 3]) and fa[:, 3].sum() > 0.5 * na
    np.testing.assert_allclose(fa[:, :3], fb[:, :3], rtol=1e-5, atol=1e-3)


def hot_flame():
    """The transcendental-free flame plus a strongly contracting xform: a ~100-pixel region that
    takes 3 % of all samples."""
    gnm, prof = linear_flame()
    gnm['xforms']['3'] = {'weight': 0.031, 'color': 0.8, 'color_speed': 0.5,
                          'pre_affine': configs._affine(15.0, 0.05, 0.3, -0.2),
                          'variations': {'linear': {'weight': 1.0}}}
    return gnm, dict(prof, width=512, height=512)


def point_flame(share):
    """The transcendental-free flame plus an xform that maps everything onto ONE point: that pixel
    takes about `share` of all samples (share 1: the point xform alone)."""
    gnm, prof = linear_flame()
    point = {'weight': 1.0, 'color': 0.8, 'color_speed': 0.5, 'pre_affine': configs._affine(0.0, 0.0, 0.21, -0.13),
             'variations': {'linear': {'weight': 1.0}}}
    if share >= 1.0:
        gnm['xforms'] = {'0': point}
    else:
        total = sum(x['weight'] for x in gnm['xforms'].values())
        gnm['xforms']['3'] = dict(point, weight=total * share / (1.0 - share))
    return gnm, dict(prof, width=512, height=512)


@pytest.mark.parametrize('share', [1.0, 0.9, 0.6, 0.3])
def test_binned_point_attractor_keeps_every_sample(mgr_prod, share):
    """A pixel that receives most of the samples: thousands of adds reach its LDS cell between the
    moment the count passes the drain threshold and the drain, enough to carry out of the 10-bit
    count (before the step-level test in k_accum_tiles sent such groups straight to the float
    accumulator, this flame lost 6-33 % of its density).  75 M samples in two launches: the float32
    accumulators are past 2^24 and no longer count exactly on either side, so the density is held to
    the number of accepted samples within float rounding instead of bit for bit."""
    gnm, prof = point_flame(share)
    res, ref_state, dev_state, dim, _ = run_device_model(mgr_prod, gnm, prof, nrounds=96, fuse=16, launches=2, mode=1)
    accepted = 0
    for k, r in enumerate(res):
        assert np.array_equal(r['ctr_dev'][:2], r['ctr_ref'][:2]), (k, r['ctr_dev'], r['ctr_ref'])
        accepted += int(r['ctr_dev'][0])
        dd, dr = r['front_dev'][:, 3].astype(np.float64), r['front_ref'][:, 3].astype(np.float64)
        assert abs(dd.sum() - accepted) <= 2e-3 * accepted, (share, k, dd.sum(), accepted)
        assert abs(dr.sum() - accepted) <= 2e-3 * accepted, (share, k, dr.sum(), accepted)
        hot = int(np.argmax(dr))
        assert dr[hot] >= 0.25 * share * accepted and abs(dd[hot] - dr[hot]) <= 2e-3 * dr[hot], (share, k, dd[hot], dr[hot])
        cold = dr < 2 ** 22                                   # everywhere else the counts are exact
        assert np.array_equal(dd[cold], dr[cold]), (share, k, int((dd[cold] != dr[cold]).sum()))
        np.testing.assert_allclose(r['front_dev'][:, :3], r['front_ref'][:, :3], rtol=1e-2, atol=1e-3)   # float32 sums of 10^7 terms, grouped differently
    assert np.array_equal(dev_state[0], ref_state[0])


def test_iter_hot_pixels_and_spill(mgr):
    """
    Hot-pixel machinery: cells that fill up are drained by the overflow path, the flush sets
    hot flags, later launches thin hot pixels by roulette.  Sample bookkeeping and density are
    integer / power-of-two arithmetic and must match the oracle exactly; colour sums may be
    regrouped by the timing-dependent drain and are compared to float tolerance.
    (A flame that puts >512 hits per round on one pixel would wrap the 10-bit counters on any
    implementation of this cell format, the reference included; see DESIGN.md.)
    """
    gnm, prof = hot_flame()
    res, ref_state, dev_state, dim, _ = run_device_model(mgr, gnm, prof, nrounds=16, fuse=16, launches=4)
    last = res[-1]
    assert (last['hot_dev'] != 0).any(), 'expected hot pixels'
    assert int(last['ctr_dev'][2]) > 0, 'expected roulette drops'
    assert sum(int(r['ctr_dev'][3]) for r in res) > 0, 'expected drains of full cells'
    for k, r in enumerate(res):
        assert np.array_equal(r['ctr_dev'][:3], r['ctr_ref'][:3]), (k, r['ctr_dev'], r['ctr_ref'])
        nd = int((r['front_dev'][:, 3] != r['front_ref'][:, 3]).sum())
        assert nd == 0, (k, nd, float(r['front_dev'][:, 3].sum()), float(r['front_ref'][:, 3].sum()))
        np.testing.assert_allclose(r['front_dev'][:, :3], r['front_ref'][:, :3], rtol=2e-6, atol=1e-4)
        assert np.array_equal(r['hot_dev'], r['hot_ref']), k
    assert np.array_equal(dev_state[0], ref_state[0])


@pytest.mark.parametrize('layout', ['narrow', 'wide'])
def test_binned_equals_atomic_equals_oracle(mgr, layout, monkeypatch):
    """(layout 'wide': the 256x64-tile variant used above 4K, forced here with FLAME_BIN_WIDE.)
    The binned accumulate (LDS tile sort -> sample log -> LDS tile atomics -> coalesced packed
    adds) produces the same packed histogram, bit for bit, as direct global atomics and as the
    oracle: integer adds commute.  Odd round counts exercise a partial last batch."""
    gnm, prof = linear_flame()
    prof = dict(prof, width=1920, height=1080)
    gnm['camera']['scale'] = 1.0          # zoomed in: every cell stays below the 128-hit packed-add limit
    if layout == 'wide':                  # environment switches are read when a context is created
        monkeypatch.setenv('FLAME_BIN_WIDE', '1')
        mgr = render.RenderManager(device=0, nslots=NSLOTS, host_seed=42)
    res_a, ref_a, dev_a, dim, seeds = run_device_model(mgr, gnm, prof, nrounds=13, fuse=5, launches=1, mode=0)
    res_b, ref_b, dev_b, dim, _ = run_device_model(mgr, gnm, prof, nrounds=13, fuse=5, launches=1, mode=1, seeds_in=seeds)
    a, b = res_a[0], res_b[0]
    assert int(a['ctr_dev'][3]) == 0
    assert np.array_equal(a['ctr_dev'][:2], b['ctr_dev'][:2])
    assert np.array_equal(a['atom_dev'], a['atom_ref'])
    assert np.array_equal(b['atom_dev'], a['atom_dev'])
    assert np.array_equal(b['front_dev'], a['front_dev'])
    assert np.array_equal(dev_a[0], dev_b[0]) and np.array_equal(dev_a[1][:, :3], dev_b[1][:, :3])
    if layout == 'wide':
        mgr.fb.free()


@pytest.mark.parametrize('nw,nslots,layout', [(8, 1024, 'narrow'), (8, 1024, 'wide'), (16, 1024, 'narrow'), (16, 1024, 'wide')])
def test_larger_workgroups_bit_exact(nw, nslots, layout, monkeypatch, built):
    """The 8-wave (from ~1440p) and 16-wave (8K) walker geometries: direct atomics, the binned
    accumulate and the oracle's device model (same geometry: nw x 64 point swap) agree bit for bit
    — packed histogram, counters, RNG states and walker points — over two launches with a partial
    last batch; then a hot flame whose cells drain in LDS and at the tile add."""
    monkeypatch.setenv('FLAME_NW', str(nw))
    if layout == 'wide':
        monkeypatch.setenv('FLAME_BIN_WIDE', '1')
    m = render.RenderManager(device=0, nslots=nslots, host_seed=43)
    assert (m.fb.nw, m.fb.nthreads) == (nw, nw * 64) and walkers(m)[3].nw == nw
    gnm, prof = linear_flame()
    prof = dict(prof, width=1920, height=1080)
    gnm['camera']['scale'] = 1.0
    res_a, ref_a, dev_a, dim, seeds = run_device_model(m, gnm, prof, nrounds=21, fuse=5, launches=2, mode=0)
    res_b, ref_b, dev_b, dim, _ = run_device_model(m, gnm, prof, nrounds=21, fuse=5, launches=2, mode=1, seeds_in=seeds)
    for k, (a, b) in enumerate(zip(res_a, res_b)):
        for r in (a, b):                      # each mode against the oracle run the same way
            assert int(r['ctr_dev'][3]) == 0 and int(r['ctr_dev'][0]) > 0
            assert np.array_equal(r['ctr_dev'][:2], r['ctr_ref'][:2])
            assert np.array_equal(r['atom_dev'], r['atom_ref'])
            assert np.array_equal(r['front_dev'][:, 3], r['front_ref'][:, 3])
        if k == 0:                            # no hot flags yet: atomics do not thin, the modes must coincide
            assert np.array_equal(b['atom_dev'], a['atom_dev']) and np.array_equal(b['front_dev'], a['front_dev'])
    for dev, ref in ((dev_a, ref_a), (dev_b, ref_b)):
        assert np.array_equal(dev[0], ref[0])                          # RNG states
        assert np.array_equal(dev[1][:, :3], ref[1][:, :3])            # walker points
    gnm, prof = hot_flame()
    res, ref_state, dev_state, dim, _ = run_device_model(m, gnm, prof, nrounds=40, fuse=16, launches=2, mode=1)
    for k, r in enumerate(res):
        assert np.array_equal(r['ctr_dev'][:2], r['ctr_ref'][:2]), (k, r['ctr_dev'], r['ctr_ref'])
        assert int((r['front_dev'][:, 3] != r['front_ref'][:, 3]).sum()) == 0
        np.testing.assert_allclose(r['front_dev'][:, :3], r['front_ref'][:, :3], rtol=1e-5, atol=1e-4)   # float atomics: order varies
    assert np.array_equal(dev_state[0], ref_state[0])
    m.fb.free()


@pytest.mark.parametrize('layout', ['narrow', 'wide'])
def test_binned_hot_region_exact_density(mgr, layout, monkeypatch):
    """Binned mode on the hot-region flame, 3 launches: drains happen in LDS and at the tile add;
    density must still be exact against the oracle run without hot-pixel thinning."""
    if layout == 'wide':
        monkeypatch.setenv('FLAME_BIN_WIDE', '1')
        mgr = render.RenderManager(device=0, nslots=NSLOTS, host_seed=42)
    gnm, prof = hot_flame()
    res, ref_state, dev_state, dim, _ = run_device_model(mgr, gnm, prof, nrounds=40, fuse=16, launches=3, mode=1)
    for k, r in enumerate(res):
        assert np.array_equal(r['ctr_dev'][:2], r['ctr_ref'][:2]), (k, r['ctr_dev'], r['ctr_ref'])
        assert int(r['ctr_dev'][2]) == 0
        nd = int((r['front_dev'][:, 3] != r['front_ref'][:, 3]).sum())
        assert nd == 0, (k, nd)
        np.testing.assert_allclose(r['front_dev'][:, :3], r['front_ref'][:, :3], rtol=2e-6, atol=1e-4)
    assert np.array_equal(dev_state[0], ref_state[0])
    if layout == 'wide':
        mgr.fb.free()


def density(front, dim):
    return front[:, 3].reshape(dim.ah, dim.astride).astype(np.float64)


@pytest.mark.parametrize('mode', [0, 1])
def test_iter_distribution_cfg2(mgr, mode):
    """
    Flame with hardware transcendentals (spherical, swirl): GPU histogram vs the flam3-style CPU
    chaos game (independent per-sample xform choice).  Criteria: relative L1 of
    the normalised density on 8x8 blocks <= 2 %, block z-scores: std < 1.3, 99.9 % < 6.5, max < 9, accepted fraction
    within 0.3 %, mean colour within 1/255.
    """
    gnm, prof = small(configs.cfg2, 480, 270, samples=2 ** 26)
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    g = rdr._handle(mgr.fb)
    mgr._copy(rdr, gnm)
    dim = mgr.fb.calc_dim(gprof.width, gprof.height)
    lib = _lib.load()
    _lib.check(lib.fl_interp(mgr.fb.ctx, g, dim.w, dim.h, 0.5, 0.0))
    run = C.c_uint64()
    _lib.check(lib.fl_iterate(mgr.fb.ctx, g, dim.w, dim.h, float(2 ** 26), 256, mode, C.byref(run)))
    nbins = dim.ah * dim.astride
    front = mgr.fb.read('front', (nbins, 4), np.float32)
    n_gpu = run.value
    F = prepare(gnm, prof)
    ref, secs, acc = O.flam3_render(F['dim'], F['packer'].prog, F['params'], F['palette'], F['seeds'], 2 ** 26, 8)
    dg, dr = density(front, dim), density(ref, dim)
    # accepted fraction
    assert abs(dg.sum() / n_gpu - dr.sum() / 2 ** 26) < 3e-3, (dg.sum() / n_gpu, dr.sum() / 2 ** 26)
    H, W = dim.ah // 8 * 8, dim.astride // 8 * 8
    bg = dg[:H, :W].reshape(H // 8, 8, W // 8, 8).sum((1, 3))
    br = dr[:H, :W].reshape(H // 8, 8, W // 8, 8).sum((1, 3))
    pg, pr = bg / bg.sum(), br / br.sum()
    l1 = np.abs(pg - pr).sum()
    assert l1 < 0.02, l1
    s = br.sum() / bg.sum()
    z = (bg * s - br) / np.sqrt(br + bg * s * s + 1.0)
    # Wave-coherent xform choice is cluster sampling: the CPU device model, in the reference's
    # own 8x32 geometry as well as in 4x64, shows z std 1.03-1.18 and maxima up to ~7 against
    # the independent flam3-style game (Poisson gives 1.0 / ~4).  Bound spread, tail, extreme.
    assert z.std() < 1.3, z.std()
    assert np.percentile(np.abs(z), 99.9) < 6.5, np.percentile(np.abs(z), 99.9)
    assert np.abs(z).max() < 9.0, np.abs(z).max()
    # colour: density-weighted mean of each channel per unit density
    cg = front[:, :3].sum(0) / dg.sum()
    cr = ref[:, :3].sum(0) / dr.sum()
    assert np.abs(cg - cr).max() < 1.0 / 255, (cg, cr)


# ---------------------------------------------------------------------------------- filters
def synth_accum(dim, seed=1):
    """A plausible accumulation buffer: smooth blobs + sparse noise + empty regions, YUV-ish."""
    rs = np.random.RandomState(seed)
    H, W = dim.ah, dim.astride
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    dens = np.zeros((H, W), np.float32)
    for _ in range(6):
        cx, cy, s = rs.uniform(0, W), rs.uniform(0, H), rs.uniform(8, 60)
        dens += rs.uniform(20, 2000) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    dens = rs.poisson(dens).astype(np.float32)
    dens[: H // 5] = 0
    buf = np.zeros((H, W, 4), np.float32)
    buf[..., 3] = dens
    buf[..., 0] = dens * rs.uniform(0.1, 0.9, (H, W))
    buf[..., 1] = dens * rs.uniform(0.3, 0.7, (H, W))
    buf[..., 2] = dens * rs.uniform(0.3, 0.7, (H, W))
    return buf.reshape(-1, 4)


def run_filter(mgr, name, dim, buf, vals):
    lib = _lib.load()
    _lib.check(lib.fl_debug_clear(mgr.fb.ctx, dim.w, dim.h, 0))
    mgr.fb.write('front', buf)
    arr = np.asarray(vals, np.float32)
    _lib.check(lib.fl_filter(mgr.fb.ctx, _lib.FILT[name], dim.w, dim.h, arr.ctypes.data, len(arr)))
    return mgr.fb.read('front', buf.shape, np.float32)


def assert_close(dev, ref, rtol, atol, what):
    assert np.isfinite(dev[np.isfinite(ref)]).all(), '%s: %d non-finite device values where the oracle is finite' % (
        what, (~np.isfinite(dev[np.isfinite(ref)])).sum())
    err = np.abs(dev - ref) - (atol + rtol * np.abs(ref))
    bad = err > 0
    assert not bad.any(), '%s: %d / %d out of tolerance, worst dev=%r ref=%r' % (
        what, bad.sum(), bad.size, dev.flat[np.argmax(err)], ref.flat[np.argmax(err)])


FW, FH = 200, 120


def test_filter_yuv(mgr):
    dim = mgr.fb.calc_dim(FW, FH); d = O.calc_dim(FW, FH)
    buf = synth_accum(dim)
    dev = run_filter(mgr, 'yuv', dim, buf, [])
    assert_close(dev, O.yuv_to_rgb(d, buf), 1e-6, 1e-6, 'yuv_to_rgb')


def test_filter_logscale(mgr):
    dim = mgr.fb.calc_dim(FW, FH); d = O.calc_dim(FW, FH)
    buf = synth_accum(dim)
    dev = run_filter(mgr, 'logscale', dim, buf, [4.1875, 0.002])
    # hardware log / rcp (reference: -use_fast_math): 1e-3 relative class, observed ~1e-6
    assert_close(dev, O.logscale(d, buf, 4.1875, 0.002), 1e-4, 1e-6, 'logscale')


def test_filter_colorclip(mgr):
    dim = mgr.fb.calc_dim(FW, FH); d = O.calc_dim(FW, FH)
    buf = O.logscale(d, O.yuv_to_rgb(d, synth_accum(dim)), 4.1875, 0.002)
    for hp in (-1.0, 2.0):
        vals = [1.0, hp, 0.25, 0.01, 0.01 ** (0.25 - 1)]
        dev = run_filter(mgr, 'colorclip', dim, buf, vals)
        assert_close(dev, O.colorclip(d, buf, *vals), 1e-3, 1e-5, 'colorclip hp=%g' % hp)


def test_filter_bilateral_chain(mgr):
    dim = mgr.fb.calc_dim(FW, FH); d = O.calc_dim(FW, FH)
    buf = O.yuv_to_rgb(d, synth_accum(dim))
    vals = [6.0 * FW / 1920., 0.05, 1.5, 0.8, 4.0]
    dev = run_filter(mgr, 'bilateral', dim, buf, vals)
    ref = O.bilateral_chain(d, buf, *vals)
    # 8 chained passes of ~31 fast-math exp/pow taps each: 1e-3 relative, 1e-4 absolute
    assert_close(dev, ref, 2e-3, 2e-4, 'bilateral chain')


@pytest.mark.parametrize('cstd', [0.02, 0.006])
def test_filter_bilateral_narrow_colour_kernel(mgr, cstd):
    """A colour standard deviation well below the default 0.05 (the profile's `color_std` is a free
    spline): cs = -17 / -57 per unit of squared colour distance, and colours above 1 after yuv -> rgb.
    The one-kernel-per-direction form once factored 2^(cs*|c|^2) out of the tap loop: the partial
    exponents overflowed and whole regions came back as 0 / 1 (tools/soak_filters.py)."""
    dim = mgr.fb.calc_dim(FW, FH); d = O.calc_dim(FW, FH)
    buf = O.yuv_to_rgb(d, synth_accum(dim, seed=5))
    n = buf[:, :3] / np.maximum(buf[:, 3:4], 1e-20)
    assert (n[buf[:, 3] > 0] ** 2).sum(1).max() > 2.0           # |c|^2 > 2: cs * |c|^2 beyond the exp2 range at cstd 0.006
    vals = [6.0 * FW / 1920., cstd, 1.5, 0.8, 4.0]
    dev = run_filter(mgr, 'bilateral', dim, buf, vals)
    ref = O.bilateral_chain(d, buf, *vals)
    assert np.isfinite(dev).all()
    err = np.abs(dev - ref)
    tol = 2e-4 + 2e-3 * np.abs(ref)
    assert (err > tol).mean() < 2e-4, ((err > tol).mean(), err.max())


# The three cases outside tools/soak_filters.py's bars in 120 (colour deviations of 0.003-0.004, a twelfth of the default).  Round 4
# put them down to cancellation in the expanded colour term; round 5 traced them (tools/diag_de_cancel.py, diag_de_cancel2.py):
# every deviating value sits at an input-EMPTY pixel, and the chain agrees with the oracle until some pass produces a pixel whose
# raw weight sums lie within a decade of FLT_MIN.  With such a narrow colour kernel every weight in the far field is ~2^-100, the
# reference divides by (weightsum + 1e-10) — a factor 1e10 for those pixels — and a density of 1e-27 comes back whose colour sums
# have lost terms to flush-to-zero (the reference runs with -ftz, cuburn/code/util.py:96): which terms depends on the order of the
# operations (factor * pix.x against (f * w) * n here: 5.7e-28 against 4.7e-28 in one channel of the first such pixel of case 94).
# That pixel is LIVE in the next pass, its colour decides its weight, 2^(-103 * cdiff), and where it is the largest term of a
# neighbour's weight sum the neighbour moves by 40 % — then its neighbours.  No operation order but the reference's own
# reproduces it (the literal per-tap form does), and the reference on its own hardware, with other exponentials, would not either.
# What IS pinned: everything that is not downstream of such a pixel.  `frontier` = pixels some pass leaves with a density in
# (0, 1e-20); a pass spreads the mark along its taps (radius 15 in its direction).
SOAK_FRONTIER_CASES = [25, 60, 94]


def _soak_case(mgr, k):
    rs = np.random.RandomState(7000 + k)
    w, h = int(rs.choice([96, 161, 320, 480, 641])), int(rs.choice([64, 97, 180, 270, 359]))
    dim = mgr.fb.set_dim(w, h); d = O.calc_dim(w, h)
    acc = (synth_accum if k % 2 == 0 else sparse_accum)(dim, seed=k + 1)
    buf = O.yuv_to_rgb(d, acc)
    bil = [float(rs.uniform(0.5, 12.0)), float(10 ** rs.uniform(-2.5, -0.3)), float(rs.uniform(0.3, 4.0)), float(rs.uniform(0.3, 1.2)), float(rs.uniform(0.5, 8.0))]
    return dim, d, buf, bil


@pytest.mark.parametrize('case', SOAK_FRONTIER_CASES)
def test_filter_bilateral_underflow_frontier(mgr, case):
    dim, d, buf, bil = _soak_case(mgr, case)
    assert bil[1] < 0.0045                                           # the narrow colour kernels of the soak
    dev = run_filter(mgr, 'bilateral', dim, buf, bil).reshape(dim.ah, dim.astride, 4)
    pat = [(1.0, 0.0), (0.0, 1.0), (1.0, 1.0), (-1.0, 1.0), (1.0, 0.5), (-0.5, 1.0), (1.0, -0.5), (0.5, 1.0)]      # cuburn/code/filters.py:8-17
    ref = buf.copy()
    mark = np.zeros((dim.ah, dim.astride), bool)
    for p in range(8):
        # a pass spreads what is marked along its taps ...
        spread = mark.copy()
        for r in range(-15, 16):
            dx, dy = int(np.rint(pat[p][0] * r)), int(np.rint(pat[p][1] * r))
            ys = np.clip(np.arange(dim.ah) + dy, 0, dim.ah - 1); xs = np.clip(np.arange(dim.astride) + dx, 0, dim.astride - 1)
            spread |= mark[np.ix_(ys, xs)]
        ref, _, _ = O.bilateral_pass(d, ref, p, *bil)
        w = ref.reshape(dim.ah, dim.astride, 4)[..., 3]
        mark = spread | ((w > 0) & (w < 1e-20))                      # ... and may leave new pixels at the underflow frontier
    ref = ref.reshape(dim.ah, dim.astride, 4)
    assert np.array_equal(ref.reshape(-1, 4), O.bilateral_chain(d, buf, *bil))
    err = np.abs(dev - ref); tol = 2e-4 + 2e-3 * np.abs(ref)
    off = (err > tol).any(-1)
    assert np.isfinite(dev).all()
    assert not (off & ~mark).any(), ((off & ~mark).sum(), err[~mark].max())      # the standard bar wherever underflow has no say
    assert off.any() and off.mean() < 0.02, (off.sum(), off.mean())               # (the cases do show the effect, on a few pixels)
    if case % 2 == 0:
        assert mark.mean() < 0.75, mark.mean()                                   # dense images: a quarter to a half of the picture is pinned by the bar above (the rest by the one below)
    # (the sparse case's frontier reaches every pixel of its 97 rows within eight passes: there the statement below is the pin)
    assert not (off & (buf.reshape(dim.ah, dim.astride, 4)[..., 3] > 0)).any()   # no pixel that held samples is affected


# All 31 taps carry weight here (at sstd = 6 * 200 / 1920 the spatial coefficient is 4e-5 at r = 3 and below
# 1e-8 from r = 4: the tests above exercise +-3 taps).  With sstd 6 / 12 / 24 — what 1080p / 4K / 8K frames
# pass to the kernel — spa_coefs[15] = exp(-225 / (sqrt2 * sstd)) = 3e-12 / 1.8e-6 / 1.3e-3: the outer taps, the
# +-16 prefetch and the half-slope rounding at large |r| all matter.  Also negative gradient speeds and the ends of
# the density-power range.
#   * dpow = 0: powf(0, 0) = 1 in the oracle (C).  The reference's fast-math powf = exp2f(0 * -inf) = NaN at every
#     empty pixel, i.e. the reference is broken there; the device defines w^0 = 1 (flame_device.h de_pow).
#   * dpow = 2 is ill-conditioned at high density whatever the implementation: the density factor is
#     exp2(-0.5 / dstd * |w_c^2 - w_q^2|), and at w = 2000 one float32 ulp of w^2 (0.25 .. 0.5) moves the exponent by
#     0.1, the tap weight by 7 % — the hardware pow (exp2(y * log2 x), ~10 ulp there; CUDA's __powf likewise) and
#     glibc's powf then legitimately differ by 10 % in the output.  The dense buffer is therefore scaled to
#     densities <= 40 for dpow = 2 (ulp(w^2) <= 1.2e-4); the default dpow = 0.8 is benign up to w ~ 1e6.
WIDE_DE = [(6.0, 0.05, 1.5, 0.8, 4.0), (12.0, 0.05, 1.5, 0.8, 4.0), (24.0, 0.05, 1.5, 0.8, 4.0),
           (24.0, 0.05, 1.5, 0.8, -3.0), (12.0, 0.1, 0.7, 0.0, 4.0), (12.0, 0.05, 3.0, 2.0, 1.0),
           (24.0, 0.02, 1.5, 0.0, -6.0), (6.0, 0.05, 1.5, 2.0, 8.0)]


@pytest.mark.parametrize('kind', ['dense', 'sparse'])
@pytest.mark.parametrize('vals', WIDE_DE)
def test_filter_bilateral_wide_parameters(mgr, kind, vals):
    dim = mgr.fb.calc_dim(FW, FH); d = O.calc_dim(FW, FH)
    buf = O.yuv_to_rgb(d, synth_accum(dim, seed=11)) if kind == 'dense' else sparse_accum(dim, seed=13)
    vals = list(vals)
    if vals[3] == 2.0:
        buf = buf * np.float32(40.0 / buf[:, 3].max())
    dev = run_filter(mgr, 'bilateral', dim, buf, vals)
    ref = O.bilateral_chain(d, buf, *vals)
    assert np.isfinite(ref).all()
    assert_close(dev, ref, 2e-3, 2e-4, 'bilateral chain %s %r' % (kind, vals))


def sparse_accum(dim, seed=3):
    """A few-samples-per-pixel buffer: isolated single hits, empty gaps, black and saturated colours
    (the regime where the DE weights underflow to denormals)."""
    rs = np.random.RandomState(seed)
    H, W = dim.ah, dim.astride
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    lam = 0.02 + 3.0 * np.exp(-((xx - W * 0.6) ** 2 + (yy - H * 0.5) ** 2) / (2 * 25.0 ** 2))
    lam[:, : W // 4] = 0.002
    dens = rs.poisson(lam).astype(np.float32)
    dens[H // 2, W // 8] = 900.0                      # one bright pixel in the empty quarter
    col = rs.choice(np.array([0.0, 0.5, 1.0], np.float32), size=(H, W, 3))
    buf = np.zeros((H, W, 4), np.float32)
    buf[..., 3] = dens
    buf[..., :3] = dens[..., None] * col
    return buf.reshape(-1, 4)


def test_filter_bilateral_sparse(built):
    """Low-density input: the DE chain stays finite and agrees with the oracle.  (Rounds 1-4 also ran the literal per-tap kernel
    and round 1's blur + packed-math pair here; round 5 removed both from the library.)"""
    form = 'one kernel per direction'
    m = render.RenderManager(device=0, nslots=NSLOTS, host_seed=7)
    dim = m.fb.calc_dim(FW, FH); d = O.calc_dim(FW, FH)
    buf = sparse_accum(dim)
    vals = [6.0 * FW / 1920., 0.05, 1.5, 0.8, 4.0]
    dev = run_filter(m, 'bilateral', dim, buf, vals)
    ref = O.bilateral_chain(d, buf, *vals)
    assert np.isfinite(ref).all()
    assert_close(dev, ref, 2e-3, 2e-4, 'sparse bilateral chain (%s)' % form)
    # energy is conserved up to the filter's own normalisation: no NaN/Inf swallowed by later clamps
    assert abs(dev[:, 3].sum() - ref[:, 3].sum()) < 1e-3 * ref[:, 3].sum()
    m.fb.free()


def test_filter_smearclip_chain(mgr):
    dim = mgr.fb.calc_dim(FW, FH); d = O.calc_dim(FW, FH)
    buf = O.logscale(d, O.yuv_to_rgb(d, synth_accum(dim)), 4.1875, 0.02)
    vals = [0.7, 0.25 - 1, 0.01, 0.01 ** (0.25 - 1)]
    dev = run_filter(mgr, 'smearclip', dim, buf, vals)
    assert_close(dev, O.smearclip_chain(d, buf, *vals), 1e-3, 1e-5, 'smearclip chain')


def test_filter_haloclip_plainclip_logencode(mgr):
    dim = mgr.fb.calc_dim(FW, FH); d = O.calc_dim(FW, FH)
    buf = O.logscale(d, O.yuv_to_rgb(d, synth_accum(dim)), 4.1875, 0.02)
    dev = run_filter(mgr, 'haloclip', dim, buf, [-0.75])
    assert_close(dev, O.haloclip_chain(d, buf, -0.75), 1e-3, 1e-5, 'haloclip')
    vals = [-0.75, 0.01, 0.01 ** -0.75, 1.3]
    dev = run_filter(mgr, 'plainclip', dim, buf, vals)
    assert_close(dev, O.plainclip(d, buf, *vals), 1e-3, 1e-5, 'plainclip')
    pos = np.maximum(buf, 1e-3)
    dev = run_filter(mgr, 'logencode', dim, pos, [2.2])
    assert_close(dev, O.logencode(d, pos, 2.2), 1e-3, 1e-4, 'logencode')


@pytest.mark.parametrize('fmt', [0, 1])
def test_output_bit_exact(mgr, fmt):
    lib = _lib.load()
    dim = mgr.fb.calc_dim(FW, FH); d = O.calc_dim(FW, FH)
    rs = np.random.RandomState(5)
    buf = rs.uniform(-0.2, 1.3, (dim.ah * dim.astride, 4)).astype(np.float32)
    _lib.check(lib.fl_debug_clear(mgr.fb.ctx, dim.w, dim.h, 0))
    mgr.fb.write('front', buf)
    seeds = mgr.fb.read('seeds', (mgr.fb.nwalkers, 3), np.uint32)
    out = np.zeros((FH, FW, 4), np.uint16 if fmt else np.uint8)
    _lib.check(lib.fl_output(mgr.fb.ctx, FW, FH, fmt, out.ctypes.data, 0))
    _lib.check(lib.fl_ctx_sync(mgr.fb.ctx))
    # the dither kernel owns the last 65536 RNG states (walkers | palette rows | output dither)
    nwalk = walkers(mgr)[2]
    ref, rng_after = O.f32_to_rgba(d, buf, seeds[nwalk + 64 * 256:], fmt)
    assert len(seeds) - (nwalk + 64 * 256) == mgr.fb.nout
    after = mgr.fb.read('seeds', (mgr.fb.nwalkers, 3), np.uint32)
    assert np.array_equal(after[nwalk + 64 * 256:], rng_after)
    assert np.array_equal(out, ref)
    # cuburn/code/tests/test_output.py ranges: negative -> 0, >1 -> peak
    peak = 65535 if fmt else 255
    assert (out[buf.reshape(dim.ah, dim.astride, 4)[12:12 + FH, 12:12 + FW] <= 0] == 0).all()
    assert (out[buf.reshape(dim.ah, dim.astride, 4)[12:12 + FH, 12:12 + FW] > 1.0] == peak).all()


@pytest.mark.parametrize('fmt', [2, 3, 4, 5])
def test_output_yuv_bit_exact(mgr, fmt):
    """Planar YUV for the video encoders (cuburn/code/output.py:75-221): every plane and the dither
    RNG states afterwards equal the oracle's, bit for bit."""
    lib = _lib.load()
    dim = mgr.fb.calc_dim(FW, FH); d = O.calc_dim(FW, FH)
    assert FW % 2 == 0 and FH % 2 == 0
    rs = np.random.RandomState(6)
    buf = rs.uniform(-0.2, 1.3, (dim.ah * dim.astride, 4)).astype(np.float32)
    buf[rs.uniform(size=len(buf)) < 0.3] = 0.0                 # empty pixels: zero alpha in the 4:2:0 weights
    _lib.check(lib.fl_debug_clear(mgr.fb.ctx, dim.w, dim.h, 0))
    mgr.fb.write('front', buf)
    seeds = mgr.fb.read('seeds', (mgr.fb.nwalkers, 3), np.uint32)
    nwalk = walkers(mgr)[2]
    ref, rng_after = O.f32_to_rgba(d, buf, seeds[nwalk + 64 * 256:], fmt)
    assert lib.fl_output_bytes(FW, FH, fmt) == ref.nbytes
    out = np.zeros_like(ref)
    _lib.check(lib.fl_output(mgr.fb.ctx, FW, FH, fmt, out.ctypes.data, 0))
    _lib.check(lib.fl_ctx_sync(mgr.fb.ctx))
    after = mgr.fb.read('seeds', (mgr.fb.nwalkers, 3), np.uint32)
    assert np.array_equal(after[nwalk + 64 * 256:], rng_after)
    assert np.array_equal(out, ref)


def test_output_yuv_reference_known_answers(mgr):
    """cuburn/code/tests/test_output.py:23-125 through the device, at the reference's 640x360."""
    lib = _lib.load()
    w, h = 640, 360
    dim = mgr.fb.set_dim(w, h)
    _lib.check(lib.fl_debug_clear(mgr.fb.ctx, w, h, 0))        # allocates the buffers for this size

    def run(fmt, pixels=(), fill=0.0):
        buf = np.full((dim.ah, dim.astride, 4), fill, np.float32)
        for (y, x), v in pixels:
            buf[12 + y, 12 + x] = v
        mgr.fb.write('front', buf.reshape(-1, 4))
        out = np.zeros(lib.fl_output_bytes(w, h, fmt), np.uint8)
        _lib.check(lib.fl_output(mgr.fb.ctx, w, h, fmt, out.ctypes.data, 0))
        _lib.check(lib.fl_ctx_sync(mgr.fb.ctx))
        return out if fmt == 2 else out.view(np.uint16)

    for fill, luma in ((-1.0, 0), (5.0, 255)):
        o = run(2, fill=fill).reshape(3, h, w)
        assert (o[0] == luma).all() and o[1].min() >= 127 and o[1].max() <= 128 and o[2].min() >= 127 and o[2].max() <= 128
    o = run(3).reshape(3, h, w)
    assert (o[0] == 0).all() and (o[1] > 510).all() and (o[1] < 513).all() and (o[2] > 510).all() and (o[2] < 513).all()
    green = [0, 1, 0, 1]
    o = run(3, [((0, 0), green), ((1, 1), green)]).reshape(3, h, w)
    assert o[0, 0, 0] > 0 and o[0, 1, 1] > 0 and o[1, 0, 0] < 500 and o[1, 1, 1] < 500
    o = run(4, [((0, 0), green), ((2, 2), green), ((3, 3), [1, 0, 0, 1])])
    luma = o[:w * h].reshape(h, w)
    cb = o[w * h:w * h + w * h // 4].reshape(h // 2, w // 2)
    assert luma[0, 0] > 0 and luma[1, 0] == 0 and luma[0, 1] == 0 and luma[1, 1] == 0 and luma[2, 2] > 0 and luma[3, 3] > 0
    assert 172 <= cb[0, 0] <= 174 and 511 <= cb[0, 1] <= 512 and 511 <= cb[1, 0] <= 512
    # odd sizes cannot be subsampled
    assert lib.fl_output(mgr.fb.ctx, 641, 360, 4, None, 0) == _lib.FL_E_INVAL
    assert lib.fl_output(mgr.fb.ctx, 640, 360, 6, None, 0) == _lib.FL_E_INVAL and lib.fl_output_bytes(640, 360, 6) == 0
    mgr.fb.set_dim(FW, FH)


def test_video_outputs_through_queue_frame(mgr, tmp_path):
    """A profile with a video output renders through queue_frame into the encoder's pixel format and
    the encoder receives exactly the frames the device produced."""
    import os, stat, sys
    from cuburn_amd import encoders
    fake = tmp_path / 'enc'
    fake.write_text('#!%s\nimport sys\nsys.stdout.buffer.write(sys.stdin.buffer.read())\n' % sys.executable)
    os.chmod(str(fake), os.stat(str(fake)).st_mode | stat.S_IXUSR)
    gnm, prof = small(configs.cfg3, 320, 180, samples=2 ** 22)
    for out, nbytes in ((encoders.VPxOutput(codec='vp9', pix_fmt='yuv420p10', command=str(fake)), 320 * 180 * 3),
                        (encoders.VPxOutput(codec='vp8', command=str(fake)), 320 * 180 * 3 // 2),
                        (encoders.X264Output(command=str(fake)), 320 * 180 * 6)):
        gprof = profile.wrap(prof, gnm)
        rdr = render.Renderer(gnm, gprof)
        rdr.out = out
        sent = 0
        for tc in (0.25, 0.5):
            evt, h_out = mgr.queue_frame(rdr, gnm, gprof, tc)
            evt.synchronize()
            assert h_out.shape == out.shape(mgr.fb.calc_dim(320, 180)) and h_out.dtype == np.dtype(out.dtype)
            assert h_out.max() > 0
            assert out.encode(h_out) == ({}, [])
            sent += 1
        media, logs = out.encode(None)
        assert len(next(iter(media.values())).read()) == sent * nbytes


def test_queue_frame_end_to_end(mgr):
    """The drop-in entry point: Renderer + RenderManager.queue_frame -> (evt, h_out)."""
    gnm, prof = small(configs.cfg3, 320, 180, samples=2 ** 24)
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    mgr.timings_reset()
    evt, h_out = mgr.queue_frame(rdr, gnm, gprof, 0.25)
    assert evt.query() in (True, False)
    evt.synchronize()
    assert evt.query() is True
    assert evt.time() > 0
    assert h_out.shape == (180, 320, 4) and h_out.dtype == np.uint8
    assert h_out[..., 3].max() > 100 and (h_out[..., :3].max() > 50)
    t = mgr.timings()
    assert t['iter_ms'] > 0 and t['filter_ms'] > 0
    media, logs = rdr.out.encode(h_out)
    assert media


# ---------------------------------------------------------------------------------- larger configs
@pytest.mark.parametrize('prod', [False, True])
def test_cfg3_animated_distribution(mgr, mgr_prod, prod):
    """cfg3: 8 xforms + final xform, two interpolated palettes, temporal sampling over the frame
    window (td > 0: one parameter block per slot, 64 palette rows) against the flam3-style game
    driven by the oracle's own parameter blocks and palette — with 1024 slots and with the
    production 1536."""
    mgr = mgr_prod if prod else mgr
    gnm, prof = small(configs.cfg3, 480, 270, samples=2 ** 26)
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    g = rdr._handle(mgr.fb)
    mgr._copy(rdr, gnm)
    dim = mgr.fb.calc_dim(gprof.width, gprof.height)
    lib = _lib.load()
    tc = 0.37
    ts, td = frame_times(gprof, tc)
    assert td > 0
    _lib.check(lib.fl_interp(mgr.fb.ctx, g, dim.w, dim.h, ts, td))
    run = C.c_uint64()
    _lib.check(lib.fl_iterate(mgr.fb.ctx, g, dim.w, dim.h, float(2 ** 26), 64, 1, C.byref(run)))
    nbins = dim.ah * dim.astride
    front = mgr.fb.read('front', (nbins, 4), np.float32)
    F = prepare(gnm, prof, tc, nslots=mgr.fb.nslots)
    ref, secs, acc = O.flam3_render(F['dim'], F['packer'].prog, F['params'], F['palette'], F['seeds'], 2 ** 26, 8)
    dg, dr = density(front, dim), density(ref, dim)
    assert abs(dg.sum() / run.value - dr.sum() / 2 ** 26) < 3e-3
    H, W = dim.ah // 8 * 8, dim.astride // 8 * 8
    bg = dg[:H, :W].reshape(H // 8, 8, W // 8, 8).sum((1, 3))
    br = dr[:H, :W].reshape(H // 8, 8, W // 8, 8).sum((1, 3))
    assert np.abs(bg / bg.sum() - br / br.sum()).sum() < 0.025
    cg = front[:, :3].sum(0) / dg.sum()
    cr = ref[:, :3].sum(0) / dr.sum()
    assert np.abs(cg - cr).max() < 1.5 / 255, (cg, cr)


def test_4k_binned_equals_atomic(mgr):
    """3840x2160 (1085 tiles of 128x64): binned and direct-atomic accumulate agree bit for bit."""
    gnm, prof = linear_flame()
    prof = dict(prof, width=3840, height=2160)
    gnm['camera']['scale'] = 0.6
    res_a, _, dev_a, dim, seeds = run_device_model_gpu_only(mgr, gnm, prof, nrounds=13, fuse=5, mode=0)
    res_b, _, dev_b, dim, _ = run_device_model_gpu_only(mgr, gnm, prof, nrounds=13, fuse=5, mode=1, seeds_in=seeds)
    assert np.array_equal(res_a['ctr'][:2], res_b['ctr'][:2])
    assert int(res_a['ctr'][3]) == 0
    assert np.array_equal(res_a['atom'], res_b['atom'])
    assert int((res_a['atom'] >> np.uint64(54)).sum()) == int(res_a['ctr'][0])
    assert np.array_equal(dev_a, dev_b)


@pytest.mark.parametrize('nw,size', [(8, (3840, 2160)), (16, (3840, 2160)), (16, (7680, 4320))])
def test_large_images_large_workgroups_binned_equals_atomic(nw, size, monkeypatch, built):
    """The geometries that ship above 1440p — 8- and 16-wave workgroups with the tile-count scan
    shared by all waves (more than 512 tiles), narrow tiles at 4K and 256x64 tiles at 8K — against
    direct atomics on the same device: packed histograms, counters and RNG states bit for bit."""
    monkeypatch.setenv('FLAME_NW', str(nw))
    m = render.RenderManager(device=0, nslots=1024, host_seed=44)
    assert m.fb.nw == nw
    gnm, prof = linear_flame()
    prof = dict(prof, width=size[0], height=size[1])
    gnm['camera']['scale'] = 0.6
    res_a, _, dev_a, dim, seeds = run_device_model_gpu_only(m, gnm, prof, nrounds=19, fuse=5, mode=0)
    res_b, _, dev_b, dim, _ = run_device_model_gpu_only(m, gnm, prof, nrounds=19, fuse=5, mode=1, seeds_in=seeds)
    assert ((dim.astride + 127) // 128) * ((dim.ah + 63) // 64) > 512
    assert np.array_equal(res_a['ctr'][:2], res_b['ctr'][:2]) and int(res_a['ctr'][3]) == 0
    assert np.array_equal(res_a['atom'], res_b['atom'])
    assert int((res_a['atom'] >> np.uint64(54)).sum()) == int(res_a['ctr'][0]) > 0
    assert np.array_equal(dev_a, dev_b)
    m.fb.free()


def test_long_launch_log_beyond_4gb_binned_equals_atomic(monkeypatch, built):
    """A frame of more than 1024 rounds follows the reference's growing batches when that saves a launch (flame_abi.hip:
    FL_BIN_MAX_ROUNDS_LONG): a launch of 1536 rounds of the 8K geometry writes a sample log of 6.4 GB and a directory of 98 304
    batches per tile — byte offsets past 2^32, batch numbers past 2^16.  One such launch, binned against direct atomics on the same
    walkers and seeds: cells drain into the float accumulator on both sides, so the flushed density is compared — whole numbers
    below 2^24, equal cell for cell — together with the counters and the RNG states."""
    monkeypatch.setenv('FLAME_NW', '16')
    m = render.RenderManager(device=0, nslots=1024, host_seed=45)
    lib = _lib.load()
    gnm, prof = linear_flame()
    prof = dict(prof, width=7680, height=4320)
    gnm['camera']['scale'] = 0.6
    rdr, gprof, dim, g, ts, td = setup_frame(m, gnm, prof)
    nbins = dim.ah * dim.astride
    seeds0 = m.fb.read('seeds', (m.fb.nwalkers, 3), np.uint32)
    nrounds, fuse = 1536, 5
    assert nrounds * 1024 * 16 * 64 * 4 > 2 ** 32
    out = {}
    for mode in (0, 1):
        m.fb.write('seeds', seeds0)
        _lib.check(lib.fl_debug_clear(m.fb.ctx, dim.w, dim.h, 1))
        _lib.check(lib.fl_debug_iter_launch(m.fb.ctx, g, dim.w, dim.h, 0, nrounds + fuse, fuse, mode))
        _lib.check(lib.fl_debug_flush(m.fb.ctx, dim.w, dim.h))
        ctr = np.zeros(4, np.uint64)
        _lib.check(lib.fl_debug_counters(m.fb.ctx, ctr.ctypes.data))
        dens = m.fb.read('front', (nbins, 4), np.float32)[:, 3].copy()
        out[mode] = (ctr, dens, m.fb.read('seeds', (m.fb.nwalkers, 3), np.uint32)[:walkers(m)[2]])
    (ca, da, ra), (cb, db, rb) = out[0], out[1]
    assert np.array_equal(ca[:3], cb[:3]) and int(ca[2]) == 0            # same accepted / out-of-frame counts, nothing thinned
    assert int(ca[0]) > 0.3 * 1024 * 1024 * nrounds
    assert da.max() < 2 ** 24 and np.array_equal(da, db)
    assert int(da.astype(np.float64).sum()) == int(ca[0])
    assert np.array_equal(ra, rb)
    m.fb.free()


def animated_cfg5():
    """cfg5's twelve heavy xforms (more than the per-genome kernel keeps resident: records fetched per round, operand table in LDS)
    with a moving xform, a moving offset and a turning camera, so that every temporal sample has its own parameter block."""
    gnm, prof = configs.cfg5(samples=2 ** 26)
    gnm['time'] = {'duration': 1, 'frame_width': 1.0}
    gnm['camera'] = dict(gnm['camera'], rotation=[0.0, 15.0, 15.0, 15.0])
    gnm['xforms']['03']['pre_affine']['angle'] = [30.0, 50.0, 80.0, 50.0]
    gnm['xforms']['07']['pre_affine']['offset']['x'] = [-0.3, 0.5, 0.2, 0.5]
    return gnm, dict(prof, frame_width=1.0, fps=24, duration=2)


@pytest.mark.parametrize('rtc', ['1', '0'])
@pytest.mark.parametrize('geom', [(8, 512), (16, 256)])
@pytest.mark.parametrize('which,size', [('cfg3', (640, 360)), ('cfg3', (3840, 2160)), ('cfg5', (1280, 720))])
def test_paired_halves_are_the_walkers_of_1024_four_wave_slots(which, size, geom, rtc, monkeypatch, built):
    """512 slots of 8 waves whose halves — and 256 slots of 16 waves whose quarters — walk their own temporal samples (render.py
    WIDE_FEW / HUGE_FEW, iter.hip "Sub-blocks of four waves") are, walker for walker, the 1024 slots of 4 waves of the reference's geometry: same seeds, same parameter block per walker (an ANIMATED genome:
    every temporal sample has its own block), same point swap inside each half.  Counters, RNG states and walker points must agree
    bit for bit, and so must the flushed density (cells of these flames fill up and drain into the float accumulator in an order
    that is not reproducible even between two runs of ONE geometry, so the packed cells themselves are not compared; the colour
    sums agree to float rounding) — direct atomics and the binned accumulate (whose batches now hold two samples' records), the
    per-genome kernel and the interpreter."""
    monkeypatch.setenv('FLAME_RTC', rtc)
    gnm, prof = small(configs.cfg3, size[0], size[1]) if which == 'cfg3' else animated_cfg5()
    prof = dict(prof, width=size[0], height=size[1])
    out = {}
    for tag, nw, nslots in (('four', 4, 1024), ('paired',) + geom):
        if nw != 4:
            monkeypatch.setenv('FLAME_NW', str(nw))
        else:
            monkeypatch.delenv('FLAME_NW', raising=False)
        m = render.RenderManager(device=0, nslots=nslots, host_seed=46)
        assert (m.fb.nw, m.fb.nslots, m.fb.nwalkers) == (nw, nslots, 1024 * 256 + 64 * 256 + 65536)
        seeds = None
        for mode in (0, 1):                                  # the binned run from the same RNG states as the direct one
            res, _, rng, dim, seeds = run_device_model_gpu_only(m, gnm, prof, nrounds=21, fuse=5, mode=mode, seeds_in=seeds)
            pts = m.fb.read('points', (1024 * 256, 4), np.float32)
            _lib.check(_lib.load().fl_debug_flush(m.fb.ctx, dim.w, dim.h))
            front = m.fb.read('front', (dim.ah * dim.astride, 4), np.float32)
            out[tag, mode] = (res['ctr'].copy(), front, rng.copy(), pts.view(np.uint32).copy())
        m.fb.free()
    for mode in (0, 1):
        (ca, fa, ra, pa), (cb, fb, rb, pb) = out['four', mode], out['paired', mode]
        assert int(ca[0]) > 0.2 * 1024 * 256 * 21, ca
        assert np.array_equal(ca[:3], cb[:3]), (mode, ca, cb)
        assert np.array_equal(ra, rb) and np.array_equal(pa, pb), mode
        assert np.array_equal(fa[:, 3], fb[:, 3]) and int(fa[:, 3].astype(np.float64).sum()) == int(ca[0]), mode
        np.testing.assert_allclose(fa[:, :3], fb[:, :3], rtol=1e-5, atol=1e-4)
    # binned == atomic in the paired geometry itself
    assert np.array_equal(out['paired', 0][1][:, 3], out['paired', 1][1][:, 3])


def run_device_model_gpu_only(mgr, gnm, prof, nrounds, fuse, mode, seeds_in=None):
    lib = _lib.load()
    if seeds_in is not None:
        mgr.fb.write('seeds', seeds_in)
    seeds0 = mgr.fb.read('seeds', (mgr.fb.nwalkers, 3), np.uint32)
    rdr, gprof, dim, g, ts, td = setup_frame(mgr, gnm, prof)
    nbins = dim.ah * dim.astride
    _lib.check(lib.fl_debug_clear(mgr.fb.ctx, dim.w, dim.h, 1))
    _lib.check(lib.fl_debug_iter_launch(mgr.fb.ctx, g, dim.w, dim.h, 0, nrounds + fuse, fuse, mode))
    ctr = np.zeros(4, np.uint64)
    _lib.check(lib.fl_debug_counters(mgr.fb.ctx, ctr.ctypes.data))
    atom = mgr.fb.read('atom', (nbins,), np.uint64)
    rng = mgr.fb.read('seeds', (mgr.fb.nwalkers, 3), np.uint32)[:walkers(mgr)[2]]
    return dict(ctr=ctr, atom=atom), None, rng, dim, seeds0


def test_8k_wide_binned_equals_atomic(mgr):
    """7680x4320 has 4148 tiles of 128x64 (> 2047): the binned accumulate switches to 256x64 tiles
    with separately staged tile numbers (2108 tiles) and still agrees with direct atomics bit for bit."""
    gnm, prof = linear_flame()
    prof = dict(prof, width=7680, height=4320)
    gnm['camera']['scale'] = 0.6
    res_a, _, dev_a, dim, seeds = run_device_model_gpu_only(mgr, gnm, prof, nrounds=19, fuse=5, mode=0)
    res_b, _, dev_b, dim, _ = run_device_model_gpu_only(mgr, gnm, prof, nrounds=19, fuse=5, mode=1, seeds_in=seeds)
    assert ((dim.astride + 127) // 128) * ((dim.ah + 63) // 64) > 2047
    assert np.array_equal(res_a['ctr'][:2], res_b['ctr'][:2])
    assert int(res_a['ctr'][0]) > 0.3 * NSLOTS * 256 * 19
    assert int(res_a['ctr'][3]) == 0
    assert np.array_equal(res_a['atom'], res_b['atom'])
    assert int((res_a['atom'] >> np.uint64(54)).sum()) == int(res_a['ctr'][0])
    assert np.array_equal(dev_a, dev_b)
    assert render.RenderManager.resolve_accum_mode(mgr, dim) == _lib.ACCUM_BINNED


# ---------------------------------------------------------------------------------- sample sharding
def test_sample_sharded_frame_never_waits_on_the_host(built):
    """queue_frame_sharded's body (distributed.sharded_frame_steps) for TWO VIRTUAL RANKS of one process: two managers with
    the ranks' seeds, the collectives stood in for by device operations on torch's stream (sum of the two accumulators cut
    into the band + halos; concatenation of the ranks' rows).  The native lanes and torch's stream are ordered by events only
    (fl_stream_dependency): when the second rank's call returns, the GPU still has most of the frame's work in front of it
    — a host wait anywhere between fl_iterate and the return would have drained it (round 4 had four).  And the frame is the
    frame: bands, halos and rows land where an unsharded render puts them."""
    import time
    import torch
    from cuburn_amd import distributed as D
    gnm, prof = configs.cfg2(samples=2 ** 31)                     # 2^30 samples per rank: ~10 ms of GPU work for the two of them
    gprof = profile.wrap(prof, gnm)
    tc = 0.5
    world = 2
    mgrs = [render.RenderManager(device=0, host_seed=D.rank_seed(42, r)) for r in range(world)]
    rdrs = [render.Renderer(gnm, gprof) for _ in range(world)]

    def frame():
        gens = [D.sharded_frame_steps(mgrs[r], rdrs[r], gnm, gprof, tc, r, world, device=0) for r in range(world)]
        reqs = [next(g) for g in gens]                            # both ranks: interp + iterate queued, at the exchange
        assert all(q[0] == 'exchange' for q in reqs)
        plan = reqs[0][2]
        rows_per, bands = plan
        total = reqs[0][1] + reqs[1][1]                           # stand-in for the reduce-scatter's sum
        ah = total.shape[0]
        replies = []
        for r in range(world):
            hp = D.halo_plan(plan, r, ah)
            r0, r1 = bands[r]
            replies.append((total[r0 - hp['top']:r1 + hp['bot']].clone(), hp['top']))
        reqs = [g.send(rep) for g, rep in zip(gens, replies)]     # both ranks: band filtered + converted, at the gather
        assert all(q[0] == 'gather' for q in reqs)
        allb = torch.cat([q[1] for q in reqs])
        done = []
        for g in gens:
            try:
                g.send(allb)
                raise AssertionError('the generator should have finished')
            except StopIteration as fin:
                done.append(fin.value)
        return done

    for _ in range(2):                                            # warm up: per-genome kernel, buffers, pinned memory
        for evt, _h in frame():
            evt.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = frame()
    t_host = time.perf_counter() - t0
    busy = [not evt.query() for evt, _h in done]
    for evt, _h in done:
        evt.synchronize()
    t_all = time.perf_counter() - t0
    assert all(busy), 'the frames had finished when queue_frame_sharded returned: something waited on the host'
    assert t_host < 0.6 * t_all, (t_host, t_all)
    ph = done[0][0].phases()
    assert set(ph) == {'iterate', 'exchange', 'filter', 'gather'} and all(v >= 0 for v in ph.values())
    # every virtual rank holds the same finished frame, and it is the unsharded frame up to sampling noise
    a, b = np.array(done[0][1]), np.array(done[1][1])
    assert np.array_equal(a, b) and a.shape == (1080, 1920, 4) and a[..., :3].max() > 50
    whole = []
    for _ in range(2):
        evt, h = mgrs[1].queue_frame(rdrs[1], gnm, gprof, tc)
        evt.synchronize()
        whole.append(np.array(h)[..., :3].astype(np.float64))
    floor = np.abs(whole[0] - whole[1]).mean()
    diff = max(np.abs(a[..., :3] - whole[0]).mean(), np.abs(a[..., :3] - whole[1]).mean())
    assert diff < 1.25 * floor + 0.25, (diff, floor)
    for m in mgrs:
        m.fb.free()


def test_sample_sharded_frames_allocate_nothing_after_warm_up(built):
    """The band path of a sample-sharded frame (distributed.sharded_frame_steps + exchange_bands' buffers) re-uses its device
    tensors — band + halos, 8-bit band, gathered frame, conversion target — and runs the reduce-scatter on the accumulator where
    it lies (fl_reserve: rows the ranks do not divide get room behind the frame's own rows instead of a zero-padded copy per
    frame).  Twenty frames of two virtual ranks after two warm-up frames: torch's allocator is not asked for a single block
    inside the library's steps (the stand-ins for the collectives, which are the test's own, are not counted), and the padded
    view handed to the exchange aliases the accumulator."""
    import torch
    from cuburn_amd import distributed as D
    gnm, prof = configs.cfg2(samples=2 ** 26)
    prof = dict(prof, width=480, height=720)                      # ah = 752: two bands of 384 rows = 768, sixteen rows of reserve
    gprof = profile.wrap(prof, gnm)
    tc, world = 0.5, 2
    mgrs = [render.RenderManager(device=0, host_seed=D.rank_seed(42, r)) for r in range(world)]
    rdrs = [render.Renderer(gnm, gprof) for _ in range(world)]
    dim = mgrs[0].fb.calc_dim(gprof.width, gprof.height)
    plan = D.band_plan(dim.ah, world)
    assert plan is not None
    rows_per = plan[0]
    nalloc = lambda: torch.cuda.memory_stats(0)['allocation.all.allocated']
    counted = [0]

    def step(fn, *a):
        before = nalloc()
        r = fn(*a)
        counted[0] += nalloc() - before
        return r

    saw_padded = []

    def frame():
        gens = [D.sharded_frame_steps(mgrs[r], rdrs[r], gnm, gprof, tc, r, world, device=0) for r in range(world)]
        reqs = [step(next, g) for g in gens]
        assert all(q[0] == 'exchange' for q in reqs)
        for q in reqs:
            if q[3] is not None:
                assert q[3].data_ptr() == q[1].data_ptr() and q[3].shape[0] == rows_per * world
                saw_padded.append(True)
        total = reqs[0][1] + reqs[1][1]
        replies = []
        for r in range(world):
            hp = D.halo_plan(plan, r, dim.ah)
            r0, r1 = plan[1][r]
            replies.append((total[r0 - hp['top']:r1 + hp['bot']].clone(), hp['top']))
        reqs = [step(g.send, rep) for g, rep in zip(gens, replies)]
        allb = torch.cat([q[1] for q in reqs])
        done = []
        for g in gens:
            try:
                step(g.send, allb)
                raise AssertionError('the generator should have finished')
            except StopIteration as fin:
                done.append(fin.value)
        return done

    for _ in range(2):
        for evt, _h in frame():
            evt.synchronize()
    counted[0] = 0
    last = None
    for _ in range(20):
        last = frame()
    for evt, _h in last:
        evt.synchronize()
    assert counted[0] == 0, '%d device allocations inside the sharded frame steps of 20 frames' % counted[0]
    if rows_per * world != dim.ah:
        assert saw_padded, 'the accumulator was not handed out with its reserve'
    a = np.array(last[0][1])
    assert a.shape == (720, 480, 4) and a[..., :3].max() > 50
    for m in mgrs:
        m.fb.free()
    D.ShardBuffers.clear()


def test_sample_sharded_frame_two_virtual_ranks(built):
    """a frame split by samples.  Two contexts with the per-rank seeds each iterate
    their share; the accumulators are summed through the zero-copy torch views that the RCCL
    all-reduce uses (here a device add stands in for the collective: one GPU); the filter chain
    runs on the sum.  The result must match the unsharded frame statistically, and the views must
    alias the native buffers exactly."""
    import torch
    from cuburn_amd import distributed as D
    gnm, prof = small(configs.cfg2, 480, 270, samples=2 ** 25)
    gprof = profile.wrap(prof, gnm)
    lib = _lib.load()
    tc = 0.5
    ts, td = frame_times(gprof, tc)
    total = int(gprof.spp(tc) * gprof.width * gprof.height)
    shares = [D.sample_share(total, r, 2) for r in range(2)]
    assert sum(shares) == total
    mgrs = [render.RenderManager(device=0, nslots=NSLOTS, host_seed=D.rank_seed(42, r)) for r in range(2)]
    rdrs = [render.Renderer(gnm, gprof) for _ in range(2)]
    dim = mgrs[0].fb.calc_dim(gprof.width, gprof.height)
    nbins = dim.ah * dim.astride
    accs, ran = [], []
    for m, rd, n in zip(mgrs, rdrs, shares):
        fid = C.c_uint32()
        _lib.check(lib.fl_frame_begin(m.fb.ctx, C.byref(fid)))
        m._copy(rd, gnm)
        g = rd._handle(m.fb)
        _lib.check(lib.fl_interp(m.fb.ctx, g, dim.w, dim.h, ts, td))
        run = C.c_uint64()
        _lib.check(lib.fl_iterate(m.fb.ctx, g, dim.w, dim.h, float(n), m.fuse, m.resolve_accum_mode(dim), C.byref(run)))
        ran.append(run.value)
        accs.append(D.accumulator_tensor(m.fb, 0))
    host = [m.fb.read('front', (nbins, 4), np.float32) for m in mgrs]
    for t, h in zip(accs, host):
        assert t.numel() == nbins * 4
        assert np.array_equal(t.cpu().numpy().reshape(nbins, 4), h)       # the view aliases the buffer
    assert not np.array_equal(host[0], host[1])                          # different RNG streams
    accs[0] += accs[1]                                                    # stand-in for all_reduce(SUM)
    torch.cuda.synchronize()
    summed = mgrs[0].fb.read('front', (nbins, 4), np.float32)
    assert np.array_equal(summed, host[0] + host[1])
    # every in-frame sample is one density count (hot-pixel roulette is unbiased, not exact)
    assert abs(summed[:, 3].sum() / sum(ran) - host[0][:, 3].sum() / ran[0]) < 5e-3
    # filter chain + output on the sum, against an unsharded render of the same frame
    for filt in rdrs[0].filts:
        filt.apply(mgrs[0].fb, gprof, getattr(gprof.filters, filt.name), dim, tc)
    rdrs[0].out.convert(mgrs[0].fb, gprof, dim)
    h_out = rdrs[0].out.copy(mgrs[0].fb, dim)          # asynchronous copy into pinned memory
    _lib.check(lib.fl_ctx_sync(mgrs[0].fb.ctx))
    sharded = np.array(h_out)
    whole = []
    for _ in range(2):                                  # two unsharded renders: the noise floor
        evt, h = mgrs[1].queue_frame(rdrs[1], gnm, gprof, tc)
        evt.synchronize()
        whole.append(np.array(h)[..., :3].astype(np.float64))
    a = sharded[..., :3].astype(np.float64)
    assert a.max() > 50
    floor = np.abs(whole[0] - whole[1]).mean()
    diff = max(np.abs(a - whole[0]).mean(), np.abs(a - whole[1]).mean())
    assert diff < 1.25 * floor + 0.25, (diff, floor)
    assert abs(a.mean() - whole[0].mean()) < 1.0
    for m in mgrs:
        m.fb.free()


@pytest.mark.parametrize('world,order', [(2, None), (4, None), (3, ['bilateral', 'logscale', 'haloclip', 'smearclip'])])
def test_band_filtering_matches_whole_frame(built, world, order):
    """Sample-sharded frames filter by row bands (distributed.py): a band of summed accumulator rows plus 224 halo
    rows on either side is filtered as an image of its own, with the FULL frame's scalars.  On one GPU: cut the
    accumulator of a 1080p frame into the bands `world` ranks would own, filter each band, stitch the bands' own rows
    together, and compare with the chain run on the whole frame.  The filters are local and the halo exceeds the
    chain's reach (8 x 24 rows), so the interior is the same computation; tiles that touch a band's artificial edge
    take the nested form of the density blurs where the whole frame takes the regrouped 19-tap form (de.hip): 1e-7
    relative in the blurred density, far below the bar.  The third case adds the two other filters that look at
    neighbouring rows (haloclip: two 7-tap blurs, smearclip: four; distributed.FILTER_REACH): the chain's reach is
    then 207 of the 224 halo rows."""
    import torch
    from cuburn_amd import distributed as D
    gnm, prof = configs.cfg2()
    prof = dict(prof, spp=2 ** 26 / (1920.0 * 1080.0))
    if order is not None:
        prof = dict(prof, filter_order=order)
    gprof = profile.wrap(prof, gnm)
    m = render.RenderManager(device=0, host_seed=9)
    rdr = render.Renderer(gnm, gprof)
    lib = _lib.load()
    tc = 0.5
    dim = m.fb.set_dim(gprof.width, gprof.height)
    ts, td = frame_times(gprof, tc)
    fid = C.c_uint32()
    _lib.check(lib.fl_frame_begin(m.fb.ctx, C.byref(fid)))
    m._copy(rdr, gnm)
    g = rdr._handle(m.fb)
    _lib.check(lib.fl_interp(m.fb.ctx, g, dim.w, dim.h, ts, td))
    run = C.c_uint64()
    _lib.check(lib.fl_iterate(m.fb.ctx, g, dim.w, dim.h, float(2 ** 26), m.fuse, m.resolve_accum_mode(dim), C.byref(run)))
    acc = m.fb.read('front', (dim.ah, dim.astride * 4), np.float32)
    for filt in rdr.filts:
        filt.apply(m.fb, gprof, getattr(gprof.filters, filt.name), dim, tc)
    whole = m.fb.read('front', (dim.ah, dim.astride * 4), np.float32)
    assert whole.max() > 0.5

    assert D.band_path_ok(rdr.out, dim, [f.name for f in rdr.filts])
    plan = D.band_plan(dim.ah, world)
    assert plan is not None and all(r0 % 16 == 0 and r1 % 16 == 0 for r0, r1 in plan[1])
    assert plan[1][0][0] == 0 and plan[1][-1][1] == dim.ah and all(a[1] == b[0] for a, b in zip(plan[1], plan[1][1:]))
    stitched = np.zeros_like(whole)
    for r0, r1 in plan[1]:
        top = D.BAND_HALO if r0 > 0 else 0
        band = torch.from_numpy(acc[r0 - top:min(r1 + D.BAND_HALO, dim.ah)].copy()).cuda()
        _, bdim = D.filter_band(m, rdr, gprof, dim, band, tc, 0, convert=False)
        res = m.fb.read('front', (bdim.ah, dim.astride * 4), np.float32)
        stitched[r0:r1] = res[top:top + (r1 - r0)]
    err = np.abs(stitched - whole)
    assert err.max() < 2e-5 and err.mean() < 1e-7, (err.max(), err.mean())
    assert D.band_plan(dim.ah, 8) is None and D.band_plan(4352, 8)[0] == 544      # 1080p bands of 8 ranks are shorter than their halo; 8K: 544 rows
    m.fb.free()


def test_flam3_xml_to_frame(mgr, tmp_path):
    """Front end to pixels: flam3 XML -> node -> looping animation (genome.store) -> queue_frame."""
    import json
    from cuburn_amd.genome import store
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'genome_front.json')))
    src = gold['xml']['rich'].replace(' chaos="1 0.5 2"', '')
    path = tmp_path / 'rich.flam3'
    path.write_text(src)
    with pytest.warns(UserWarning):
        gnm, base = store.connect(str(tmp_path)).animation(str(path))      # two flames in the file: first is used
    assert base == 'rich' and gnm['type'] == 'animation'
    prof = dict(configs.cfg2()[1], width=320, height=240)
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    frames = []
    for tc in (0.1, 0.6):
        evt, h = mgr.queue_frame(rdr, gnm, gprof, tc)
        evt.synchronize()
        frames.append(np.array(h))
    for f in frames:
        assert f.shape == (240, 320, 4) and np.isfinite(f.astype(np.float32)).all()
        assert (f[..., 3] > 0).mean() > 0.05 and f[..., :3].max() > 60
    assert np.abs(frames[0].astype(np.int32) - frames[1].astype(np.int32)).mean() > 0.5     # the loop moves


@pytest.mark.parametrize('codec,suffix', [('png', '.png'), ('tiff', '.tiff'), ('jpeg', '.jpg')])
def test_cli_renders_still(built, tmp_path, codec, suffix):
    """python -m cuburn_amd FLAME --still --codec ...: the reference's main.py flow end to end."""
    import json, subprocess, sys
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'genome_front.json')))
    (tmp_path / 'B.json').write_text(json.dumps(gold['db']['B']))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-m', 'cuburn_amd', 'B', '-d', str(tmp_path), '--still', '-P', 'preview',
                          '--codec', codec, '-o', str(tmp_path), '--width', '320', '--height', '180', '--spp', '200'],
                         cwd=repo, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    files = sorted(f for f in os.listdir(str(tmp_path)) if f.endswith(suffix))
    assert len(files) == 1, (os.listdir(str(tmp_path)), out.stderr[-500:])
    data = open(os.path.join(str(tmp_path), files[0]), 'rb').read()
    assert len(data) > 2000
    if codec != 'tiff':
        PIL = pytest.importorskip('PIL.Image')
        img = np.array(PIL.open(os.path.join(str(tmp_path), files[0])))
        assert img.shape[:2] == (180, 320) and img.max() > 60
    else:
        assert data[:4] == b'II*\x00'
    assert 'ms' in out.stderr


# ---------------------------------------------------------------------------------- BASELINE full size
def test_cfg2_full_size_iterate_and_filter_chain(built):
    """
    BASELINE configs[1] at its real size (1920x1080, 2^28 samples) in the production
    configuration (default slots, binned accumulate), through size-independent properties and the
    oracle:
      * the histogram is additive and every sample has weight 1: the density channel holds
        integers, their sum cannot exceed the samples run, and the in-frame fraction agrees with
        the flam3-style CPU game run at the same 2^28 samples;
      * 8x8-block density distribution vs the CPU game (same bars as the reduced-size test);
      * two frames with different RNG states are equal within shot noise (no state leaks from
        frame to frame) and differ (the RNG does advance);
      * the full filter chain (yuv -> bilateral -> logscale -> colorclip) of the full-size
        accumulator agrees with the oracle's chain run on the same buffer, element-wise.
    """
    lib = _lib.load()
    gnm, prof = configs.cfg2()
    gprof = profile.wrap(prof, gnm)
    m = render.RenderManager(device=0, host_seed=42)              # production defaults
    rdr = render.Renderer(gnm, gprof)
    # the geometry queue_frame picks for this frame (1024 slots of 4 waves for up to 2^28 samples): what bench.py's
    # headline line runs
    dim = m.fb.set_dim(gprof.width, gprof.height, nsamples=gprof.spp(0.5) * gprof.width * gprof.height)
    assert (dim.w, dim.h) == (1920, 1080) and (m.fb.nw, m.fb.nslots) == (4, 1024)
    nbins = dim.ah * dim.astride
    tc = 0.5
    ts, td = frame_times(gprof, tc)
    g = rdr._handle(m.fb)
    fronts, runs = [], []
    for _ in range(2):
        fid = C.c_uint32()
        _lib.check(lib.fl_frame_begin(m.fb.ctx, C.byref(fid)))
        m._copy(rdr, gnm)
        _lib.check(lib.fl_interp(m.fb.ctx, g, dim.w, dim.h, ts, td))
        run = C.c_uint64()
        _lib.check(lib.fl_iterate(m.fb.ctx, g, dim.w, dim.h, float(2 ** 28), m.fuse, m.resolve_accum_mode(dim), C.byref(run)))
        assert m.resolve_accum_mode(dim) == _lib.ACCUM_BINNED
        fronts.append(m.fb.read('front', (nbins, 4), np.float32))
        runs.append(run.value)
    a, b = fronts
    assert runs[0] >= 2 ** 28 and runs[0] - 2 ** 28 < m.fb.nslots * 256
    da, db = density(a, dim), density(b, dim)
    assert np.array_equal(da, np.rint(da)) and da.min() >= 0            # integer counts
    assert da.sum() <= runs[0] and db.sum() <= runs[1]
    assert not np.array_equal(da, db)
    F = prepare(gnm, prof, tc, nslots=m.fb.nslots)
    ref, secs, acc = O.flam3_render(F['dim'], F['packer'].prog, F['params'], F['palette'], F['seeds'], 2 ** 28, 16)
    dr = density(ref, dim)
    assert abs(da.sum() / runs[0] - dr.sum() / 2 ** 28) < 2e-3, (da.sum() / runs[0], dr.sum() / 2 ** 28)
    H, W = dim.ah // 8 * 8, dim.astride // 8 * 8
    def blocks(dd):
        return dd[:H, :W].reshape(H // 8, 8, W // 8, 8).sum((1, 3))
    ba, bb, br = blocks(da), blocks(db), blocks(dr)
    assert np.abs(ba / ba.sum() - br / br.sum()).sum() < 0.02
    s = br.sum() / ba.sum()
    z = (ba * s - br) / np.sqrt(br + ba * s * s + 1.0)
    assert z.std() < 1.3 and np.percentile(np.abs(z), 99.9) < 6.5 and np.abs(z).max() < 9.0, (z.std(), np.abs(z).max())
    z2 = (ba - bb) / np.sqrt(ba + bb + 1.0)                              # frame vs frame: same estimator twice
    assert z2.std() < 1.3 and np.abs(z2).max() < 9.0, (z2.std(), np.abs(z2).max())
    ca, cr = a[:, :3].sum(0) / da.sum(), ref[:, :3].sum(0) / dr.sum()
    assert np.abs(ca - cr).max() < 1.0 / 255

    # full-size filter chain on frame b's accumulator (it is still the current front buffer)
    d = O.calc_dim(gprof.width, gprof.height)
    cur = b.copy()
    vals = {}
    for filt in rdr.filts:
        vals[filt.name] = [float(v) for v in filt.scalars(gprof, getattr(gprof.filters, filt.name), dim, tc)]
        filt.apply(m.fb, gprof, getattr(gprof.filters, filt.name), dim, tc)
    dev = m.fb.read('front', (nbins, 4), np.float32)
    assert [f.name for f in rdr.filts] == ['yuv', 'bilateral', 'logscale', 'colorclip']
    cur = O.yuv_to_rgb(d, cur)
    cur = O.bilateral_chain(d, cur, *vals['bilateral'])
    cur = O.logscale(d, cur, *vals['logscale'])
    cur = O.colorclip(d, cur, *vals['colorclip'])
    assert np.isfinite(dev).all()
    # tone-mapped values live in [0, 1]: 8 DE passes + log + gamma in fast math on both sides
    from test_gpu_fullsize import check_chain_error
    check_chain_error(np.abs(dev - cur), 'cfg2 whole frame')
    # queue_frame renders the same frame in the same context (no geometry switch), and its 8-bit frame is the chain's
    # output: alpha > 0 exactly where the tone-mapped density is visible
    gen = m.fb.generation
    evt, h_out = m.queue_frame(rdr, gnm, gprof, tc)
    evt.synchronize()
    assert m.fb.generation == gen and (m.fb.nw, m.fb.nslots) == (4, 1024)
    frame = np.array(h_out)
    vis = dev.reshape(dim.ah, dim.astride, 4)[12:12 + dim.h, 12:12 + dim.w, 3] > 0.5 / 255
    assert frame.shape == (1080, 1920, 4) and abs((frame[..., 3] > 0).mean() - vis.mean()) < 0.01
    m.fb.free()


def test_deferred_filter_fusion_is_bit_identical(mgr):
    """fl_filter defers `yuv` and the DE's un-normalising pass so that bilateral / logscale /
    colorclip can take them along in one kernel.  Looking at the buffer between the calls forces
    every step to run on its own: both ways must give the same bits, for every chain shape."""
    lib = _lib.load()
    dim = mgr.fb.calc_dim(FW, FH)
    buf = synth_accum(dim)
    steps = {'yuv': [], 'bilateral': [6.0 * FW / 1920., 0.05, 1.5, 0.8, 4.0], 'logscale': [4.1875, 0.002],
             'colorclip': [1.0, -1.0, 0.25, 0.01, 0.01 ** (0.25 - 1)], 'smearclip': [0.7, 0.25 - 1, 0.01, 0.01 ** (0.25 - 1)]}

    def run(chain, peek):
        _lib.check(lib.fl_debug_clear(mgr.fb.ctx, dim.w, dim.h, 0))
        mgr.fb.write('front', buf)
        for name in chain:
            arr = np.asarray(steps[name], np.float32)
            _lib.check(lib.fl_filter(mgr.fb.ctx, _lib.FILT[name], dim.w, dim.h, arr.ctypes.data, len(arr)))
            if peek:
                mgr.fb.read('front', buf.shape, np.float32)
        return mgr.fb.read('front', buf.shape, np.float32)

    for chain in (['yuv', 'bilateral', 'logscale', 'colorclip'], ['yuv', 'bilateral', 'logscale', 'smearclip'],
                  ['yuv', 'bilateral'], ['bilateral', 'colorclip'], ['yuv', 'logscale', 'colorclip'],
                  ['yuv', 'bilateral', 'bilateral', 'logscale'], ['yuv', 'yuv'], ['bilateral', 'logscale', 'logscale', 'colorclip']):
        a, b = run(chain, False), run(chain, True)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), chain
        assert np.isfinite(a).all()




def test_filters_match_reference_kernel_vectors(mgr):
    """The HIP filters against vectors computed by the REFERENCE's own CUDA kernel text (tests/golden/filters.npz, made by
    tests/golden/make_golden_filters.py: the kernels compiled as host C++ behind a CUDA stand-in), without the oracle in between:
    YUV -> RGB, log scale, colour clip, plain clip and the eight-direction density-estimation chain on a 64 x 48 accumulator with
    empty cells and four decades of density, at 640 sampled positions.  Tolerances as against the oracle (which equals these
    vectors to the last bit on the CPU): the device evaluates exp / log / pow / rcp with single hardware instructions."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'filters.npz'))
    w, h = int(g['width']), int(g['height'])
    dim = mgr.fb.calc_dim(w, h)
    assert (dim.astride, dim.ah) == (int(g['astride']), int(g['aheight']))
    img, a, pos = np.ascontiguousarray(g['image'].reshape(-1, 4)), g['args'], g['positions']
    dev = run_filter(mgr, 'yuv', dim, img, [])
    assert_close(dev[pos], g['out_yuv_to_rgb'], 1e-6, 1e-6, 'yuv_to_rgb vs reference kernel')
    dev = run_filter(mgr, 'logscale', dim, img, [a[0], a[1]])
    assert_close(dev[pos], g['out_logscale'], 1e-4, 1e-6, 'logscale vs reference kernel')
    dev = run_filter(mgr, 'colorclip', dim, img, [a[11], a[12], a[8], a[9], a[10]])
    assert_close(dev[pos], g['out_colorclip'], 1e-3, 1e-5, 'colorclip vs reference kernel')
    dev = run_filter(mgr, 'bilateral', dim, img, list(a[3:8]))
    assert_close(dev[pos], g['out_bilateral_chain'], 2e-3, 2e-4, 'bilateral chain vs reference kernel')


@pytest.mark.parametrize('cfg', ['cfg3', 'cfg5', 'allvars'])
def test_interp_params_match_reference_interp_kernel(mgr, cfg):
    """``fl_interp``'s parameter blocks against the blocks the REFERENCE's own generated ``interp_iter_params`` kernel wrote
    (tests/golden/interp_params.npz, tests/golden/make_golden_interp.py), by field name, without the oracle in between: device
    splines (linear and magnitude domain), camera, affine, density and variation precalc for 12 temporal samples.  2e-5 relative
    (+ the pixel-scale camera offsets' 1e-5 absolute), the bar of the oracle comparison above."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'interp_params.npz'))
    gnm, prof = configs.allvars() if cfg == 'allvars' else configs.CONFIGS[cfg]()
    w, h = [int(v) for v in gold[cfg + '_dim'][:2]]
    gprof = profile.wrap(dict(prof, width=w, height=h), gnm)
    rdr = render.Renderer(gnm, gprof)
    g = rdr._handle(mgr.fb)
    mgr._copy(rdr, gnm)
    dim = mgr.fb.calc_dim(w, h)
    times = gold[cfg + '_times']
    tstep = np.float32(times[1] - times[0])
    ns = mgr.fb.nslots
    _lib.check(_lib.load().fl_interp(mgr.fb.ctx, g, dim.w, dim.h, float(times[0]), float(np.float32(tstep * np.float32(ns)))))
    dev = mgr.fb.read('params', (ns, rdr.packer.pstride), np.float32, g)
    names = ['.'.join(n) for n in rdr.packer.packed]
    blocks = gold[cfg + '_blocks']
    for j, rn in enumerate(str(x) for x in gold[cfg + '_names']):
        key = 'den.' + rn[4:] if rn.startswith('den_') else rn
        k = names.index(key)
        got, want = dev[:len(times), k].astype(np.float64), blocks[:, j].astype(np.float64)
        err = np.abs(got - want) / (np.abs(want) + 1.0)
        assert err.max() < 2e-5, (cfg, rn, got[np.argmax(err)], want[np.argmax(err)])


@pytest.mark.parametrize('cfg', ['cfg2', 'cfg3', 'allvars'])
def test_palette_matches_reference_kernel(mgr, cfg):
    """The packed palette and the RNG states behind it against the REFERENCE's own ``interp_palette_flat`` kernel
    (tests/golden/interp_palette.npz), bit for bit (sha256 of the 64 x 256 cells and of the 16384 states after; four rows in full),
    without the oracle in between."""
    import hashlib, os
    from cuburn_amd import mwc as my_mwc
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'interp_palette.npz'))
    gnm, prof = configs.allvars() if cfg == 'allvars' else small(configs.CONFIGS[cfg], 640, 360)
    ns, nt, nwalk, _ = walkers(mgr)
    seeds = mgr.fb.read('seeds', (nwalk + 64 * 256, 3), np.uint32)
    seeds[nwalk:] = my_mwc.make_seeds(64 * 256, int(gold['host_seed']))
    mgr.fb.write('seeds', seeds)
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    g = rdr._handle(mgr.fb)
    mgr._copy(rdr, gnm)
    dim = mgr.fb.calc_dim(gprof.width, gprof.height)
    ts, td = [float(v) for v in gold[cfg + '_ts_td']]
    _lib.check(_lib.load().fl_interp(mgr.fb.ctx, g, dim.w, dim.h, ts, td))
    dev = mgr.fb.read('palette', (64, 256), np.uint64)
    np.testing.assert_array_equal(dev[[0, 1, 31, 63]], gold[cfg + '_packed_rows'])
    assert hashlib.sha256(np.ascontiguousarray(dev).tobytes()).hexdigest() == str(gold[cfg + '_packed_sha256'])
    after = mgr.fb.read('seeds', (nwalk + 64 * 256, 3), np.uint32)[nwalk:]
    assert hashlib.sha256(np.ascontiguousarray(after).tobytes()).hexdigest() == str(gold[cfg + '_rng_after_sha256'])


def test_flush_matches_reference_ptx(mgr):
    """``k_flush`` against the REFERENCE's own flush_atom PTX (tests/golden/ptx_cells.npz: the packed cells and the float
    accumulator after 6000 adds by the reference's PTX, then its flush, all through tests/golden/ptx_mini.py), without the oracle
    in between: the float accumulator after the flush is equal to the last bit, the hot flags agree cell for cell up to the
    reference's exchanged codes 1 and 2 (see tests/test_cpu_golden.py::test_packed_cell_add_and_flush_match_reference_ptx)."""
    import os
    lib = _lib.load()
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ptx_cells.npz'))
    dim = mgr.fb.calc_dim(36, 20)
    S, AH = int(gold['astride']), int(gold['aheight'])
    assert (dim.astride, dim.ah) == (S, AH)
    ncell = S * AH
    _lib.check(lib.fl_debug_clear(mgr.fb.ctx, dim.w, dim.h, 1))
    _lib.check(lib.fl_debug_clear_hot(mgr.fb.ctx, dim.w, dim.h))
    mgr.fb.write('atom', np.ascontiguousarray(gold['atom_after_add'], np.uint64))
    mgr.fb.write('front', np.ascontiguousarray(gold['out_after_add'], np.float32))
    hot = np.zeros((ncell + 15) // 16, np.uint32)
    for cell, f in gold['hot_cells']:
        hot[cell >> 4] |= np.uint32(int(f) << ((int(cell) & 15) << 1))
    mgr.fb.write('hot', hot)
    _lib.check(lib.fl_debug_flush(mgr.fb.ctx, dim.w, dim.h))
    out = mgr.fb.read('front', (ncell, 4), np.float32)
    np.testing.assert_array_equal(out, gold['out_after_flush'])
    assert not mgr.fb.read('atom', (ncell,), np.uint64).any()
    hot2 = mgr.fb.read('hot', hot.shape, np.uint32)
    flags = np.array([(int(hot2[c >> 4]) >> ((c & 15) << 1)) & 3 for c in range(ncell)], np.uint8)
    np.testing.assert_array_equal(np.array([0, 2, 1, 3], np.uint8)[flags], gold['flags_after_flush'])


@pytest.mark.parametrize('cfg,mode', [('cfg2', 'binned'), ('cfg3', 'binned'), ('cfg5', 'binned'), ('cfg2', 'atomic')])
def test_histogram_matches_reference_iterate_kernel(mgr_prod, cfg, mode):
    """The GPU's histogram (the default binned path — and, for cfg2, the direct packed atomics, the reference's own scheme — at the
    production walker geometry) against the histogram the REFERENCE's own
    ``iter`` kernel computed when run on the host (tests/golden/iter_hist.npz, tests/golden/make_golden_iter.py: a block's threads
    as coroutines, 134 M samples), without the oracle in between: cfg2's, cfg3's and cfg5's flames at 320 x 180 over the same frame
    window.  Different random streams: the fraction of samples in frame within 0.3 %, the density over 8 x 8 blocks within 2 % L1
    (wave-coherent xform choice is cluster sampling in both; the shot-noise floor of the two samples is 0.4 %), the mean colour
    within 1 / 255."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'iter_hist.npz'))
    mgr = mgr_prod
    w, h = [int(v) for v in gold[cfg + '_size']]
    gnm, prof = small(configs.CONFIGS[cfg], w, h, samples=2 ** 27)
    rdr, gprof, dim, g, ts, td = setup_frame(mgr, gnm, prof, 0.5)
    lib = _lib.load()
    run = C.c_uint64()
    _lib.check(lib.fl_iterate(mgr.fb.ctx, g, dim.w, dim.h, float(2 ** 27), 256, _lib.ACCUM_BINNED if mode == 'binned' else _lib.ACCUM_ATOMIC, C.byref(run)))
    front = mgr.fb.read('front', (dim.ah * dim.astride, 4), np.float32)
    n_ref, plotted, _ = [int(v) for v in gold[cfg + '_counts']]
    H, W = dim.ah // 8 * 8, dim.astride // 8 * 8
    b = front.reshape(dim.ah, dim.astride, 4).astype(np.float64)[:H, :W].reshape(H // 8, 8, W // 8, 8, 4).sum((1, 3))
    a = gold[cfg + '_blocks8'].astype(np.float64)
    assert a.shape == b.shape
    assert abs(plotted / n_ref - front[:, 3].astype(np.float64).sum() / run.value) < 3e-3
    pa, pb = a[..., 3] / a[..., 3].sum(), b[..., 3] / b[..., 3].sum()
    assert np.abs(pa - pb).sum() < 0.02, float(np.abs(pa - pb).sum())
    ca, cb = a[..., :3].sum((0, 1)) / a[..., 3].sum(), b[..., :3].sum((0, 1)) / b[..., 3].sum()
    assert np.abs(ca - cb).max() < 1.0 / 255, (ca, cb)
