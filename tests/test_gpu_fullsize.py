"""
BASELINE configs[2..4] at their real sizes, each with ITS OWN genome, in the production
configuration (RenderManager() defaults, binned accumulate, per-genome kernel):

  cfg3  1920x1080, 8 xforms + final xform, two interpolated palettes, frame_width > 0, 2^30 samples
  cfg4  3840x2160 animation frame (the cfg3 genome in motion), frame_width 1, 2^28 samples per frame
  cfg5  7680x4320, 12 heavy-variation xforms, 2^32 samples

For each: the density channel holds integers and conserves samples; in-frame fraction, block
distribution and mean colour agree with the flam3-style CPU game (oracle) run on the oracle's OWN
parameter blocks at a reduced sample count; and the full filter chain of the device accumulator
agrees with the oracle's chain — on the whole buffer at 1080p, on an interior window at 4K / 8K
(the filters are local: a window plus a margin wider than 8 passes x 16 taps reproduces the
interior exactly, and keeps the single-threaded CPU chain affordable).
"""
import ctypes as C

import numpy as np
import pytest

from common import O, prepare, frame_times
from cuburn_amd import configs, profile, render, _lib

pytestmark = pytest.mark.gpu


def iterate_frame(m, gnm, prof, tc, nsamples, geometry_by_samples=False):
    lib = _lib.load()
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    dim = m.fb.set_dim(gprof.width, gprof.height, nsamples=nsamples if geometry_by_samples else None)
    g = rdr._handle(m.fb)
    ts, td = frame_times(gprof, tc)
    fid = C.c_uint32()
    _lib.check(lib.fl_frame_begin(m.fb.ctx, C.byref(fid)))
    m._copy(rdr, gnm)
    _lib.check(lib.fl_interp(m.fb.ctx, g, dim.w, dim.h, ts, td))
    run = C.c_uint64()
    mode = m.resolve_accum_mode(dim)
    assert mode == _lib.ACCUM_BINNED
    _lib.check(lib.fl_iterate(m.fb.ctx, g, dim.w, dim.h, float(nsamples), m.fuse, mode, C.byref(run)))
    front = m.fb.read('front', (dim.ah * dim.astride, 4), np.float32)
    return rdr, gprof, dim, td, run.value, front


def blocks(dens, bs):
    H, W = dens.shape[0] // bs * bs, dens.shape[1] // bs * bs
    return dens[:H, :W].reshape(H // bs, bs, W // bs, bs).sum((1, 3))


def check_against_cpu_game(gnm, prof, tc, nslots, dim, front, nrun, ncpu, bs, l1_max, frac_tol, col_tol, nthreads=16):
    dens = front[:, 3].reshape(dim.ah, dim.astride).astype(np.float64)
    assert np.array_equal(dens, np.rint(dens)) and dens.min() >= 0, 'density must hold integer counts'
    assert dens.sum() <= nrun
    F = prepare(gnm, prof, tc, nslots=nslots)
    ref, secs, acc = O.flam3_render(F['dim'], F['packer'].prog, F['params'], F['palette'], F['seeds'], ncpu, nthreads)    # one private float4 histogram per thread
    dr = ref[:, 3].reshape(dim.ah, dim.astride).astype(np.float64)
    fg, fr = dens.sum() / nrun, dr.sum() / ncpu
    assert abs(fg - fr) < frac_tol, ('in-frame fraction', fg, fr)
    bg, br = blocks(dens, bs), blocks(dr, bs)
    l1 = np.abs(bg / bg.sum() - br / br.sum()).sum()
    # shot noise of the smaller (CPU) sample alone contributes ~ sum_b sqrt(2 p_b / (pi n)) to the L1
    p = br / br.sum()
    noise = np.sqrt(2.0 * p / (np.pi * max(dr.sum(), 1.0))).sum()
    assert l1 < l1_max + 1.5 * noise, ('block L1', l1, 'shot-noise floor', noise)
    cg = front[:, :3].sum(0, dtype=np.float64) / dens.sum()
    cr = ref[:, :3].sum(0, dtype=np.float64) / dr.sum()
    assert np.abs(cg - cr).max() < col_tol, ('mean colour', cg, cr)
    return l1, noise, fg, fr


# Bars of the full-size filter chain (tone-mapped values in [0, 1]).  Measured (gpurun_out/fullsize_chain_errors.txt,
# tools/diag_chain_error.py): cfg2 / cfg3 whole 1080p frames max 2.2e-5 / 2.3e-5, 99.9 % below 1.2e-6, mean 4e-8; the
# 4K / 8K windows max 1.7e-6 / 6.4e-6.  The maximum sits at single-hit pixels at the rim of the flame, where
# colorclip's linear segment below `gamma_threshold` multiplies differences by lin^(gamma - 1) = 31.6; above a DE
# density of 10 the relative error of the DE output itself is 3e-6.  The bars are ~4x the worst measured values
# (round 2 asked for max < 2e-2): hardware exp / log / rcp on one side, libm on the other.
CHAIN_MAX, CHAIN_P999, CHAIN_MEAN = 1e-4, 5e-6, 5e-7


def check_chain_error(err, what):
    stats = (float(err.max()), float(np.percentile(err, 99.9)), float(err.mean()))
    try:
        import os
        if os.path.isdir('gpurun_out'):
            with open('gpurun_out/fullsize_chain_errors.txt', 'a') as fp:
                fp.write('%s: max %.3e p99.9 %.3e mean %.3e\n' % ((what,) + stats))
    except OSError:
        pass
    assert stats[0] < CHAIN_MAX and stats[1] < CHAIN_P999 and stats[2] < CHAIN_MEAN, (what,) + stats
    return stats


def filter_chain_on_device(m, rdr, gprof, dim, tc):
    vals = {}
    for filt in rdr.filts:
        vals[filt.name] = [float(v) for v in filt.scalars(gprof, getattr(gprof.filters, filt.name), dim, tc)]
        filt.apply(m.fb, gprof, getattr(gprof.filters, filt.name), dim, tc)
    assert [f.name for f in rdr.filts] == ['yuv', 'bilateral', 'logscale', 'colorclip']
    dev = m.fb.read('front', (dim.ah * dim.astride, 4), np.float32)
    assert np.isfinite(dev).all()
    return vals, dev


def oracle_chain(d, buf, vals):
    cur = O.yuv_to_rgb(d, np.ascontiguousarray(buf))
    cur = O.bilateral_chain(d, cur, *vals['bilateral'])
    cur = O.logscale(d, cur, *vals['logscale'])
    return O.colorclip(d, cur, *vals['colorclip'])


def check_window(dim, accum, dev, vals, x0, y0, AW, AH, margin):
    """Oracle chain on the window [y0, y0+AH) x [x0, x0+AW) of the accumulator, compared with the
    device's full-image result on the window's interior (margin > the chain's reach: 8 passes of
    16 taps plus their 9-tap density blurs)."""
    assert AW % 32 == 0 and AH % 16 == 0 and x0 + AW <= dim.astride and y0 + AH <= dim.ah
    d = O.calc_dim(AW - 24, AH - 24)
    assert (d.astride, d.ah) == (AW, AH)
    acc2 = accum.reshape(dim.ah, dim.astride, 4)[y0:y0 + AH, x0:x0 + AW].reshape(-1, 4)
    ref = oracle_chain(d, acc2, vals).reshape(AH, AW, 4)[margin:AH - margin, margin:AW - margin]
    got = dev.reshape(dim.ah, dim.astride, 4)[y0 + margin:y0 + AH - margin, x0 + margin:x0 + AW - margin]
    err = np.abs(got - ref)
    assert ref[..., 3].max() > 0.2, 'the window must contain part of the flame'
    return check_chain_error(err, 'window %dx%d at (%d, %d) of %dx%d' % (AW, AH, x0, y0, dim.w, dim.h))[0]


def densest_window(dim, accum, AW, AH):
    """Top-left corner (multiples of 32 / 16) of the AW x AH window with the most samples."""
    dens = accum[:, 3].reshape(dim.ah, dim.astride)
    bs = blocks(dens.astype(np.float64), 32)
    ny, nx = AH // 32, AW // 32
    c = np.cumsum(np.cumsum(np.pad(bs, ((1, 0), (1, 0))), 0), 1)
    tot = c[ny:, nx:] - c[:-ny, nx:] - c[ny:, :-nx] + c[:-ny, :-nx]
    j, i = np.unravel_index(np.argmax(tot), tot.shape)
    return int(i) * 32, int(j) * 32


def test_cfg3_full_size(built):
    gnm, prof = configs.cfg3()
    m = render.RenderManager(device=0, host_seed=42)
    rdr, gprof, dim, td, nrun, front = iterate_frame(m, gnm, prof, 0.37, 2 ** 30, geometry_by_samples=True)
    assert (dim.w, dim.h) == (1920, 1080) and td > 0 and m.fb.nslots == 1280          # (2^30 samples: the geometry queue_frame picks, five workgroups per CU)
    assert nrun >= 2 ** 30 and nrun - 2 ** 30 < m.fb.nslots * 256
    check_against_cpu_game(gnm, prof, 0.37, m.fb.nslots, dim, front, nrun, 2 ** 27, 16, 0.02, 2e-3, 1.5 / 255)
    vals, dev = filter_chain_on_device(m, rdr, gprof, dim, 0.37)
    d = O.calc_dim(gprof.width, gprof.height)
    ref = oracle_chain(d, front, vals)
    check_chain_error(np.abs(dev - ref), 'cfg3 whole frame')
    m.fb.free()


def test_cfg4_full_size_animation_frame(built):
    """One frame of the 60-frame 4K animation (frame 23 of 60: tc from the profile's own
    enumerate_times), temporally sampled over its frame window."""
    gnm, prof = configs.cfg4()
    gprof = profile.wrap(prof, gnm)
    times = profile.enumerate_times(gprof)
    assert len(times) == 60
    tc = float(times[22][1][0])
    m = render.RenderManager(device=0, host_seed=42)
    rdr, gprof, dim, td, nrun, front = iterate_frame(m, gnm, prof, tc, 2 ** 28)
    assert (dim.w, dim.h) == (3840, 2160) and abs(td - 1.0 / 60) < 1e-9
    assert (m.fb.nw, m.fb.nslots) == (8, 1024)                         # the 8-wave geometry from ~1440p up
    check_against_cpu_game(gnm, prof, tc, m.fb.nslots, dim, front, nrun, 2 ** 27, 32, 0.02, 2e-3, 1.5 / 255)
    vals, dev = filter_chain_on_device(m, rdr, gprof, dim, tc)
    x0, y0 = densest_window(dim, front, 1024, 768)
    check_window(dim, front, dev, vals, x0, y0, 1024, 768, 224)
    # the frame through the drop-in entry point, and the next frame of the animation differs
    outs = []
    for k in (22, 23):
        evt, h_out = m.queue_frame(rdr, gnm, gprof, float(times[k][1][0]))
        evt.synchronize()
        outs.append(np.array(h_out))
    assert outs[0].shape == (2160, 3840, 4) and outs[0][..., 3].max() > 100
    assert np.abs(outs[0].astype(np.int32) - outs[1].astype(np.int32)).mean() > 0.05
    # queue_frame knows the frame's sample count: 2^28 samples take 16-wave workgroups in quarters (round 6: 256 workgroups, the
    # reference's 1024 temporal samples x 256 walkers, sort batches of 16384 records), and the frame it renders is the frame the
    # 1024-slot geometry above rendered
    assert (m.fb.nw, m.fb.nslots, m.fb.ntemporal) == (16, 256, 1024)
    rdr2, gprof2, dim2, td2, nrun2, front2 = iterate_frame(m, gnm, prof, tc, 2 ** 28)
    assert (m.fb.nw, m.fb.nslots) == (16, 256) and nrun2 == nrun
    check_against_cpu_game(gnm, prof, tc, m.fb.ntemporal, dim2, front2, nrun2, 2 ** 27, 32, 0.02, 2e-3, 1.5 / 255)
    d1, d2 = front[:, 3].astype(np.float64), front2[:, 3].astype(np.float64)
    assert abs(d1.sum() - d2.sum()) < 2e-3 * d1.sum()
    m.fb.free()


def test_cfg5_full_size(built):
    gnm, prof = configs.cfg5()
    m = render.RenderManager(device=0, host_seed=42)
    rdr, gprof, dim, td, nrun, front = iterate_frame(m, gnm, prof, 0.5, 2 ** 32)
    assert (dim.w, dim.h) == (7680, 4320) and td == 0 and (m.fb.nw, m.fb.nslots) == (16, 1024)      # the 16-wave geometry above 4K
    assert nrun >= 2 ** 32 and nrun - 2 ** 32 < m.fb.nslots * 1024
    check_against_cpu_game(gnm, prof, 0.5, m.fb.nslots, dim, front, nrun, 2 ** 28, 64, 0.02, 2e-3, 1.5 / 255, nthreads=4)   # 537 MB per thread
    vals, dev = filter_chain_on_device(m, rdr, gprof, dim, 0.5)
    x0, y0 = densest_window(dim, front, 1024, 768)
    check_window(dim, front, dev, vals, x0, y0, 1024, 768, 224)
    m.fb.free()
