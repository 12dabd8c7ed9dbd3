"""
Edge cases of the hot path through the C ABI: smallest and ragged images, sample counts below one
round, one xform and the maximum of 64, argument validation, the default filter chain, and state
that must survive a change of image size.  The xform-choice tests use linear-only flames so that
histograms are bit-exact against the oracle's device model.
"""
import ctypes as C
import os

import numpy as np
import pytest

from common import O, prepare, frame_times
from cuburn_amd import configs, profile, render, _lib
from cuburn_amd.packer import GenomePacker
from test_gpu_parity import run_device_model, NSLOTS            # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def mgr():
    from __graft_entry__ import build
    build()
    return render.RenderManager(device=0, nslots=NSLOTS, host_seed=11)


def nxf_flame(n, w=256, h=144):
    """n linear xforms on a ring, unequal weights (the density table has n-1 entries)."""
    gnm, prof = configs.cfg1()
    xfs = {}
    for i in range(n):
        a = 2 * np.pi * i / max(n, 1)
        xfs[str(i)] = {'weight': 0.2 + (i % 5) * 0.3, 'color': i / max(n - 1, 1), 'color_speed': 0.5,
                       'pre_affine': configs._affine(7.0 * i, 0.45, 0.55 * float(np.cos(a)), 0.55 * float(np.sin(a))),
                       'variations': {'linear': {'weight': 1.0}}}
    gnm['xforms'] = xfs
    return gnm, dict(prof, width=w, height=h)


@pytest.mark.parametrize('n', [1, 2, 9, 33, 64])
def test_xform_counts_bit_exact(mgr, n):
    """1 xform (empty density table), more than 8 (the density table spans several lanes' worth of
    the vector compare) and the maximum of 64: packed histogram, counters, RNG and walkers equal
    the oracle's device model."""
    gnm, prof = nxf_flame(n)
    # One or two contracting maps are a point / a Cantor dust: a single cell takes tens of
    # thousands of hits per round, which wraps the 10-bit count of the packed-atomic scheme (the
    # reference's own limitation, iter.py:361-406) in an order-dependent way.  The binned
    # accumulate is exact there; the atomic mode is compared from 9 xforms up.
    for mode in ((1,) if n <= 2 else (0, 1)):
        res, ref_state, dev_state, dim, _ = run_device_model(mgr, gnm, prof, nrounds=6, fuse=3, launches=1, mode=mode)
        r = res[0]
        assert np.array_equal(r['ctr_dev'][:3], r['ctr_ref'][:3]), (n, mode, r['ctr_dev'], r['ctr_ref'])
        assert int(r['ctr_dev'][0]) > 0
        if int(r['ctr_dev'][3]) == 0 and int(r['ctr_ref'][3]) == 0:
            assert np.array_equal(r['atom_dev'], r['atom_ref']), (n, mode)
        # (one contracting xform is a single fixed point: its cell overflows and drains to the
        # float accumulator, in an order that atomics do not fix; what is exact is the density)
        assert np.array_equal(r['front_dev'][:, 3], r['front_ref'][:, 3]), (n, mode)
        np.testing.assert_allclose(r['front_dev'][:, :3], r['front_ref'][:, :3], rtol=5e-5, atol=1e-3)   # float adds of drained chunks regroup
        assert np.array_equal(dev_state[0], ref_state[0])
        assert np.array_equal(dev_state[1][:, :3], ref_state[1][:, :3])


def test_too_many_xforms_rejected():
    gnm, prof = nxf_flame(65)
    gprof = profile.wrap(prof, gnm)
    with pytest.raises((ValueError, AssertionError)):
        rdr = render.Renderer(gnm, gprof)
        m = render.RenderManager(device=0, nslots=NSLOTS, host_seed=1)
        try:
            rdr._handle(m.fb)
        finally:
            m.fb.free()


@pytest.mark.parametrize('w,h', [(1, 1), (33, 17), (8, 300), (1000, 999)])
def test_ragged_sizes_render(mgr, w, h):
    """calc_dim pads to (32, 16) multiples with a 12-pixel gutter (render.py:79-89); the whole
    pipeline must work on any size and crop exactly the requested window."""
    gnm, prof = configs.cfg2(samples=2 ** 22)
    prof = dict(prof, width=w, height=h, spp=2 ** 22 / float(w * h) if w * h > 4096 else 400.0)
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    evt, out = mgr.queue_frame(rdr, gnm, gprof, 0.5)
    evt.synchronize()
    out = np.array(out)
    assert out.shape == (h, w, 4)
    dim = mgr.fb.calc_dim(w, h)
    d = O.calc_dim(w, h)
    assert (dim.aw, dim.ah, dim.astride) == (d.aw, d.ah, d.astride)
    assert dim.astride % 32 == 0 and dim.ah % 16 == 0 and dim.aw == w + 24
    if w * h >= 33 * 17:
        assert out[..., 3].max() > 0


def test_sample_counts_below_one_round(mgr):
    """nsamples 0, 1 and one short of a round all run exactly one write round (rounds are whole:
    render.py:331-336 rounds the count up the same way)."""
    lib = _lib.load()
    gnm, prof = nxf_flame(3)
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    g = rdr._handle(mgr.fb)
    mgr._copy(rdr, gnm)
    dim = mgr.fb.calc_dim(gprof.width, gprof.height)
    _lib.check(lib.fl_interp(mgr.fb.ctx, g, dim.w, dim.h, 0.5, 0.0))
    per_round = NSLOTS * 256
    for n, want in ((0.0, per_round), (1.0, per_round), (per_round - 1.0, per_round), (per_round + 1.0, 2 * per_round)):
        for mode in (0, 1):
            run = C.c_uint64()
            _lib.check(lib.fl_iterate(mgr.fb.ctx, g, dim.w, dim.h, n, 4, mode, C.byref(run)))
            assert run.value == want, (n, mode, run.value)
            front = mgr.fb.read('front', (dim.ah * dim.astride, 4), np.float32)
            assert 0 < front[:, 3].sum() <= want


def test_argument_validation(mgr):
    lib = _lib.load()
    gnm, prof = nxf_flame(2)
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    g = rdr._handle(mgr.fb)
    run = C.c_uint64()
    assert lib.fl_iterate(mgr.fb.ctx, g, 256, 144, 1e6, 4, 7, C.byref(run)) == _lib.FL_E_INVAL        # bad mode
    assert lib.fl_iterate(None, g, 256, 144, 1e6, 4, 0, C.byref(run)) == _lib.FL_E_INVAL              # null ctx
    assert lib.fl_iterate(mgr.fb.ctx, None, 256, 144, 1e6, 4, 0, C.byref(run)) == _lib.FL_E_INVAL     # null genome
    assert b'' != lib.fl_last_error()
    vals = (C.c_float * 2)(1.0, 2.0)
    assert lib.fl_filter(mgr.fb.ctx, _lib.FILT['bilateral'], 256, 144, vals, 2) == _lib.FL_E_INVAL    # too few scalars
    assert lib.fl_filter(mgr.fb.ctx, 999, 256, 144, vals, 2) != 0
    assert lib.fl_output(mgr.fb.ctx, 256, 144, 6, None, 0) == _lib.FL_E_INVAL                          # bad pixel format
    ms = C.c_float()
    assert lib.fl_frame_ms(mgr.fb.ctx, 0xfffffff0, C.byref(ms)) == _lib.FL_E_INVAL                     # untracked frame id
    # fl_interp before any upload
    rdr2 = render.Renderer(gnm, gprof)
    m2 = render.RenderManager(device=0, nslots=NSLOTS, host_seed=3)
    g2 = rdr2._handle(m2.fb)
    assert lib.fl_interp(m2.fb.ctx, g2, 256, 144, 0.0, 0.0) == _lib.FL_E_INVAL
    m2.fb.free()
    # a corrupted program is rejected at creation
    packer = GenomePacker(gnm)
    prog = np.array(packer.prog, np.int32).copy()
    prog[0] ^= 1
    ops = np.ascontiguousarray(packer.ops_array, np.int32).reshape(-1)
    h = C.c_void_p()
    assert lib.fl_genome_create(mgr.fb.ctx, prog.ctypes.data, len(prog), ops.ctypes.data, len(ops) // 4,
                                packer.nrows, C.byref(h)) == _lib.FL_E_INVAL


def test_malformed_op_lists_rejected(mgr):
    """Every word an op writes and every spline row it reads is checked against the block / row
    table at fl_genome_create (multi-word ops: camera / affine 6 words, CDF b words, perspective 3)."""
    lib = _lib.load()
    gnm, prof = nxf_flame(3)
    packer = GenomePacker(gnm)
    prog = np.array(packer.prog, np.int32)
    good = np.ascontiguousarray(packer.ops_array, np.int32).reshape(-1, 4)
    ps, nrows = int(prog[3]), packer.nrows

    def create(ops):
        ops = np.ascontiguousarray(ops, np.int32)
        h = C.c_void_p()
        rc = lib.fl_genome_create(mgr.fb.ctx, prog.ctypes.data, len(prog), ops.ctypes.data, len(ops), nrows, C.byref(h))
        if rc == 0:
            lib.fl_genome_destroy(h)
        return rc

    assert create(good) == 0
    kinds = good[:, 0]
    cam = int(np.where(kinds == 2)[0][0]); aff = int(np.where(kinds == 3)[0][0]); cdf = int(np.where(kinds == 4)[0][0])
    for idx, col, val in ((cam, 1, ps - 3),             # camera writes 6 words: would run past the block
                          (cam, 2, nrows - 2),          # ... and reads 4 rows
                          (aff, 1, ps - 5), (aff, 2, nrows - 5),
                          (cdf, 3, 65), (cdf, 3, 0), (cdf, 2, nrows - 1), (cdf, 1, ps - 1)):
        bad = good.copy()
        bad[idx, col] = val
        assert create(bad) == _lib.FL_E_INVAL, (idx, col, val)
    extra = np.vstack([good, [[7, ps - 2, 0, 0]]])      # perspective writes 3 words
    assert create(extra) == _lib.FL_E_INVAL
    extra = np.vstack([good, [[5, 6, 0, nrows]]])       # ratio2: second row index out of range
    assert create(extra) == _lib.FL_E_INVAL


def test_out_of_memory_is_survivable(mgr):
    """cuburn/render.py:140-147: an allocation failure frees the framebuffers and re-raises; the
    manager stays usable.  A 200000 x 200000 frame (640 GB per float4 buffer) cannot be allocated:
    the C ABI reports FL_E_NOMEM (MemoryError in Python), and a normal frame renders afterwards."""
    lib = _lib.load()
    gnm, prof = configs.cfg2(samples=2 ** 22)
    small = dict(prof, width=320, height=180, spp=2 ** 22 / (320.0 * 180.0))
    gp = profile.wrap(small, gnm)
    rdr = render.Renderer(gnm, gp)
    evt, out = mgr.queue_frame(rdr, gnm, gp, 0.5); evt.synchronize()
    before = np.array(out).astype(np.float64)
    g = rdr._handle(mgr.fb)
    run = C.c_uint64()
    assert lib.fl_iterate(mgr.fb.ctx, g, 200000, 200000, 1e6, 4, 1, C.byref(run)) == _lib.FL_E_NOMEM
    assert b'allocation' in lib.fl_last_error()
    huge = profile.wrap(dict(prof, width=200000, height=200000, spp=1e-4), gnm)
    with pytest.raises(MemoryError):
        mgr.queue_frame(render.Renderer(gnm, huge), gnm, huge, 0.5)
    evt, out = mgr.queue_frame(rdr, gnm, gp, 0.5); evt.synchronize()
    after = np.array(out).astype(np.float64)
    assert after[..., 3].max() > 100 and np.abs(after - before).mean() < 6.0


def test_duration_event_outlives_a_context_switch():
    """In auto geometry a change of image size re-creates the native context; a DurationEvent of
    a frame queued before the switch is resolved first and stays readable (it used to keep a
    dangling fl_ctx*)."""
    m = render.RenderManager(device=0, host_seed=5)
    gnm, prof = configs.cfg2(samples=2 ** 24)
    small = profile.wrap(dict(prof, width=640, height=360, spp=2 ** 24 / (640.0 * 360.0)), gnm)
    big = profile.wrap(dict(prof, width=7680, height=4320, spp=2 ** 26 / (7680.0 * 4320.0)), gnm)
    rdr_s, rdr_b = render.Renderer(gnm, small), render.Renderer(gnm, big)
    evt_a, a = m.queue_frame(rdr_s, gnm, small, 0.5)         # NOT synchronised: the double-buffered loop
    gen = m.fb.generation
    evt_b, b = m.queue_frame(rdr_b, gnm, big, 0.5)           # switches to the 8-wave geometry
    assert m.fb.generation == gen + 1
    assert evt_a.query() is True and evt_a.time() > 0         # resolved before the old context went away
    assert np.array(a)[..., 3].max() > 0
    evt_b.synchronize()
    assert evt_b.time() > 0 and np.array(b)[..., 3].max() > 0
    m.fb.free()
    with pytest.raises(_lib.FlameError):
        render.DurationEvent(m.fb, 0).synchronize()           # no context at all: a clean error, no crash


def test_per_genome_kernel_cache_turnover(mgr):
    """The process keeps at most 64 compiled per-genome kernels (the reference keeps 20 modules,
    render.py:229-245).  70 different structures force the cache to be emptied; a Renderer made
    before that keeps working (its kernel is compiled again, not called through a stale handle)."""
    gnm0, prof = nxf_flame(3, 128, 72)
    prof = dict(prof, spp=40.0)
    gp0 = profile.wrap(prof, gnm0)
    first = render.Renderer(gnm0, gp0)
    evt, out = mgr.queue_frame(first, gnm0, gp0, 0.5); evt.synchronize()
    before = np.array(out).astype(np.float64)
    for n in range(1, 71):
        gnm, _ = nxf_flame(1 + n % 60, 128, 72)
        if n >= 60:                                         # more structures: a post affine on xform 0, other variations
            gnm['xforms']['0']['post_affine'] = configs._affine(3.0 * n, 0.9, 0.0, 0.0)
            gnm['xforms']['0']['variations'] = {'linear': {'weight': 0.7}, 'bent': {'weight': 0.3}} if n % 2 else {'sinusoidal': {'weight': 1.0}}
        gp = profile.wrap(prof, gnm)
        evt, out = mgr.queue_frame(render.Renderer(gnm, gp), gnm, gp, 0.5); evt.synchronize()
        assert np.array(out)[..., 3].max() > 0, n
    evt, out = mgr.queue_frame(first, gnm0, gp0, 0.5); evt.synchronize()
    after = np.array(out).astype(np.float64)
    assert after[..., 3].max() > 0 and np.abs(after - before).mean() < 8.0


def test_default_filter_chain_and_size_changes(mgr):
    """The reference's default chain is bilateral -> logscale -> smearclip (specs.py:107); render it,
    then a different size, then the first size again: buffers are re-sized, walkers / RNG persist, and
    the two same-size frames agree within frame-to-frame noise."""
    gnm, prof = configs.cfg2(samples=2 ** 24)
    prof = dict(prof, width=320, height=200, spp=2 ** 24 / (320.0 * 200.0))
    prof.pop('filter_order', None)
    gprof = profile.wrap(prof, gnm)
    assert list(gprof.filter_order) == ['bilateral', 'logscale', 'smearclip']
    rdr = render.Renderer(gnm, gprof)
    assert [f.name for f in rdr.filts] == ['yuv', 'bilateral', 'logscale', 'smearclip']
    frames = []
    for size in ((320, 200), (640, 64), (320, 200)):
        p = dict(prof, width=size[0], height=size[1])
        gp = profile.wrap(p, gnm)
        rd = render.Renderer(gnm, gp)
        evt, out = mgr.queue_frame(rd, gnm, gp, 0.5)
        evt.synchronize()
        out = np.array(out)
        assert out.shape == (size[1], size[0], 4) and out[..., 3].max() > 100
        frames.append(out.astype(np.float64))
    assert np.abs(frames[0] - frames[2]).mean() < 6.0
    assert not np.array_equal(frames[0], frames[2])


def test_walker_geometry_follows_image_size():
    """A manager built without an explicit slot count uses 4-wave slots for small images — 1024 of
    them for frames of up to 2^28 samples, 1280 up to 2^30, 1536 above, decided per frame from its sample count (the GEOMETRY of a
    frame does not depend on what the context rendered before; its RNG streams do, as in the reference) —
    from ~1440p up 1024 8-wave slots for frames of more than 2^28 samples and 256 16-wave slots in quarters (a temporal sample per
    four waves) for frames of up to 2^28 — round 6: the 4K accumulate gains more from 16384-record batches than the walk loses —,
    and 16-wave slots above 4K — 256 in quarters / 1024 — (the native context is re-created on the switch, genome handles follow)."""
    m = render.RenderManager(device=0, host_seed=5)
    assert (m.fb.nw, m.fb.nslots) == (4, 1536)
    gnm, prof = configs.cfg2(samples=2 ** 24)
    small = profile.wrap(dict(prof, width=640, height=360, spp=2 ** 24 / (640.0 * 360.0)), gnm)
    big = profile.wrap(dict(prof, width=7680, height=4320, spp=2 ** 26 / (7680.0 * 4320.0)), gnm)
    rdr_s, rdr_b = render.Renderer(gnm, small), render.Renderer(gnm, big)
    gen0 = m.fb.generation
    evt, a = m.queue_frame(rdr_s, gnm, small, 0.5); evt.synchronize()
    assert (m.fb.nw, m.fb.nslots) == (4, 1024) and m.fb.generation == gen0 + 1 and np.array(a)[..., 3].max() > 0
    gen0 += 1
    evt, a1 = m.queue_frame(rdr_s, gnm, small, 0.5); evt.synchronize()        # same class of frame: no further switch
    assert m.fb.generation == gen0
    evt, b = m.queue_frame(rdr_b, gnm, big, 0.5); evt.synchronize()
    assert (m.fb.nw, m.fb.nslots, m.fb.ntemporal) == (16, 256, 1024) and m.fb.generation == gen0 + 1      # few samples: a sample per four waves
    b = np.array(b)
    assert b.shape == (4320, 7680, 4) and b[..., 3].max() > 0
    assert m.last_nsamples % (256 * 1024) == 0
    mid = profile.wrap(dict(prof, width=3840, height=2160, spp=2 ** 25 / (3840.0 * 2160.0)), gnm)
    evt, c = m.queue_frame(render.Renderer(gnm, mid), gnm, mid, 0.5); evt.synchronize()
    # (4K with few samples: the same 16-wave quarters the 8K frame left behind — no switch)
    assert (m.fb.nw, m.fb.nslots, m.fb.ntemporal) == (16, 256, 1024) and m.fb.generation == gen0 + 1 and np.array(c)[..., 3].max() > 0
    mid_many = profile.wrap(dict(prof, width=3840, height=2160, spp=2 ** 28.5 / (3840.0 * 2160.0)), gnm)
    evt, c2 = m.queue_frame(render.Renderer(gnm, mid_many), gnm, mid_many, 0.5); evt.synchronize()
    assert (m.fb.nw, m.fb.nslots, m.fb.ntemporal) == (8, 1024, 1024) and m.fb.generation == gen0 + 2
    c, c2 = np.array(c).astype(np.float64), np.array(c2).astype(np.float64)
    assert c2[..., 3].max() > 0 and np.abs(c - c2).mean() < 8.0               # the same picture, less noise
    evt, a2 = m.queue_frame(rdr_s, gnm, small, 0.5); evt.synchronize()        # same Renderer, new context
    assert (m.fb.nw, m.fb.nslots) == (4, 1024) and m.fb.generation == gen0 + 3
    a, a2 = np.array(a).astype(np.float64), np.array(a2).astype(np.float64)
    assert np.abs(a - a2).mean() < 6.0
    # ... with RNG states of its own: the generation count is mixed into the seed (the reference's seed table lives as long as
    # its manager, render.py:95-104; round 4 re-seeded every re-created context identically)
    assert not np.array_equal(a, a2)
    # 1024 <-> 1280 slots twice on one manager: the two 1024-slot frames, and the two 1280-slot frames, differ
    many_m = profile.wrap(dict(prof, width=640, height=360, spp=2 ** 28.5 / (640.0 * 360.0)), gnm)
    rdr_m = render.Renderer(gnm, many_m)
    flips = []
    for gp, rd in ((many_m, rdr_m), (small, rdr_s), (many_m, rdr_m), (small, rdr_s)):
        evt, f = m.queue_frame(rd, gnm, gp, 0.5); evt.synchronize()
        flips.append(((m.fb.nw, m.fb.nslots), np.array(f)))
    assert [g for g, _ in flips] == [(4, 1280), (4, 1024), (4, 1280), (4, 1024)]
    assert not np.array_equal(flips[1][1], flips[3][1]) and not np.array_equal(flips[0][1], flips[2][1])
    assert np.abs(flips[1][1].astype(np.float64) - flips[3][1]).mean() < 6.0
    # many samples per frame on a small image: the 1536-slot geometry the manager starts with stays
    q = render.RenderManager(device=0, host_seed=5)
    many = profile.wrap(dict(prof, width=640, height=360, spp=2 ** 30.5 / (640.0 * 360.0)), gnm)
    evt, _ = q.queue_frame(render.Renderer(gnm, many), gnm, many, 0.5); evt.synchronize()
    assert (q.fb.nw, q.fb.nslots) == (4, 1536) and q.fb.generation == 0
    # ... and a frame of few samples after it gets the geometry it would have got as a first frame: the same pixels
    # as manager m's first frame (same host seed, same frame), whatever this context rendered before
    evt, a3 = q.queue_frame(rdr_s, gnm, small, 0.5); evt.synchronize()
    assert (q.fb.nw, q.fb.nslots) == (4, 1024) and q.fb.generation == 1
    assert np.array_equal(np.array(a3), a.astype(np.uint8))
    q.fb.free()
    # an explicit slot count pins the geometry
    p = render.RenderManager(device=0, nslots=NSLOTS, host_seed=5)
    evt, _ = p.queue_frame(render.Renderer(gnm, big), gnm, big, 0.5); evt.synchronize()
    assert (p.fb.nw, p.fb.nslots) == (4, NSLOTS)
    m.fb.free(); p.fb.free()


def test_long_run_without_timing_queries(mgr):
    """A long render that never asks for kernel timings: the per-launch timing events are capped
    (8192 pairs since the last reset), rendering goes on, and a reset re-arms the timers."""
    gnm, prof = nxf_flame(3, 64, 64)
    prof = dict(prof, spp=40.0)
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    mgr.timings_reset()
    last = None
    for k in range(1400):                       # ~7 timed launches per frame
        evt, out = mgr.queue_frame(rdr, gnm, gprof, 0.5)
        if last is not None:
            last.synchronize()
        last = evt
    last.synchronize()
    assert np.array(out)[..., 3].max() > 0
    t = mgr.timings()
    assert 0 < t['launches'] <= 8192 and t['iter_ms'] > 0
    mgr.timings_reset()
    evt, out = mgr.queue_frame(rdr, gnm, gprof, 0.5); evt.synchronize()
    t = mgr.timings()
    assert t['launches'] == 1 and t['iter_ms'] > 0 and t['filter_ms'] > 0


def test_cli_renders_stills_and_a_video_shard(tmp_path, monkeypatch, capfd):
    """python -m cuburn_amd end to end (the reference's main.py loop): PNG stills of two frames, then
    the same animation as one x264 shard through an `x264` found on PATH (a stand-in that copies its
    input): the segment holds every frame of the shard in the encoder's input format."""
    import json, stat, sys
    from cuburn_amd import __main__ as cli
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'genome_front.json')))
    (tmp_path / 'A.json').write_text(json.dumps(gold['db']['A']))
    out = tmp_path / 'out'; out.mkdir()
    common = ['A', '-d', str(tmp_path), '-o', str(out), '--width', '160', '--height', '90', '--spp', '50',
              '--duration', '1', '--fps', '4']
    assert cli.main(common + ['--codec', 'png', '--start', '1', '--end', '3']) == 0
    stills = sorted(os.listdir(str(out)))
    assert len(stills) == 2 and all(n.endswith('.png') for n in stills), stills
    assert open(str(out / stills[0]), 'rb').read(8) == b'\x89PNG\r\n\x1a\n'
    assert '(  1/  1)' in capfd.readouterr().err
    # video: one shard of 4 frames
    bindir = tmp_path / 'bin'; bindir.mkdir()
    fake = bindir / 'x264'
    fake.write_text('#!%s\nimport sys\nsys.stderr.write("fake x264\\n")\nsys.stdout.buffer.write(sys.stdin.buffer.read())\n' % sys.executable)
    os.chmod(str(fake), os.stat(str(fake)).st_mode | stat.S_IXUSR)
    monkeypatch.setenv('PATH', str(bindir) + os.pathsep + os.environ.get('PATH', ''))
    assert cli.main(common + ['--codec', 'x264', '--shard', '1', '-n', 'clip']) == 0
    segs = [n for n in os.listdir(str(out)) if n.endswith('.h264')]
    assert segs == ['clip_00001.h264']
    data = np.fromfile(str(out / segs[0]), np.uint16)
    assert data.size == 4 * 90 * 160 * 3 and data.max() > 50
    frames = data.reshape(4, 90, 160, 3)
    assert np.abs(frames[0].astype(np.int64) - frames[3].astype(np.int64)).mean() > 0.05      # the animation moves
    assert 'fake x264' in capfd.readouterr().err


def test_cli_under_a_per_gpu_launcher_shares_the_files(tmp_path, monkeypatch, capfd):
    """Two processes' worth of environment (RANK 0 / 1 of WORLD_SIZE 2, run one after the other on the
    one GPU of the test box): each renders its own half of the files, together they are complete,
    and a second pass finds nothing left to do (resume is forced on, distribute.py:160-161)."""
    import json
    from cuburn_amd import __main__ as cli
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'genome_front.json')))
    (tmp_path / 'A.json').write_text(json.dumps(gold['db']['A']))
    out = tmp_path / 'out'; out.mkdir()
    argv = ['A', '-d', str(tmp_path), '-o', str(out), '--width', '160', '--height', '90', '--spp', '20',
            '--duration', '1', '--fps', '6', '--codec', 'png']
    monkeypatch.setenv('WORLD_SIZE', '2'); monkeypatch.setenv('LOCAL_RANK', '0')
    monkeypatch.setenv('RANK', '1')
    assert cli.main(argv) == 0
    assert sorted(os.listdir(str(out))) == ['A_00002.png', 'A_00004.png', 'A_00006.png']
    monkeypatch.setenv('RANK', '0')
    assert cli.main(argv) == 0
    assert sorted(os.listdir(str(out))) == ['A_%05d.png' % k for k in range(1, 7)]
    capfd.readouterr()
    stamp = {n: os.path.getmtime(str(out / n)) for n in os.listdir(str(out))}
    assert cli.main(argv) == 0 and capfd.readouterr().err.count('ms') == 0          # nothing rendered again
    assert stamp == {n: os.path.getmtime(str(out / n)) for n in os.listdir(str(out))}
