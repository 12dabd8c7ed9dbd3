"""
CPU tests (no GPU): internal consistency of the oracle — size-independent properties that the
GPU tests rely on (conservation, linearity of the flush, filter identities), and agreement of
its two independent iterate implementations (device model vs flam3-style game).
"""
import numpy as np
import pytest

from common import O, prepare, mwc
from cuburn_amd import configs

pytestmark = pytest.mark.usefixtures('built')


def small(cfg, w, h, **kw):
    gnm, prof = cfg(**kw)
    return gnm, dict(prof, width=w, height=h)


def test_pack_unpack_roundtrip():
    rs = np.random.RandomState(0)
    for _ in range(200):
        y, u, v = rs.randint(0, 256, 3)
        hi = (1 << 22) | (int(y) << 4)
        lo = (int(u) << 18) | int(v)
        cell = 0
        n = int(rs.randint(1, 1023))
        for _ in range(n):
            cell += (hi << 32) | lo
        assert O.unpack_cell(cell).tolist() == [n * y, n * u, n * v, n]
    assert O.unpack_cell(((1 << 22 | 255 << 4) << 32 | (255 << 18) | 255) * 1023).tolist() == [255 * 1023] * 3 + [1023]


def run_model(F, geom, rounds, fuse, seeds=None):
    d = F['dim']; nb = d.ah * d.astride
    nt = geom.nw * geom.wl
    rng = (mwc.make_seeds(F['nslots'] * nt, 3) if seeds is None else seeds).copy()
    pts = np.full((F['nslots'] * nt, 4), np.nan, np.float32)
    hot = np.zeros(nb // 16, np.uint32); atom = np.zeros(nb, np.uint64); out4 = np.zeros((nb, 4), np.float32)
    # one parameter block per slot (stills: all blocks are equal)
    ctr = O.iter_launch(geom, d, F['packer'].prog, F['params'][:F['nslots']], F['palette'], rng, pts, F['nslots'], hot, atom, out4,
                        0, rounds + fuse, fuse)
    return ctr, atom, out4, hot, rng, pts


def test_device_model_conservation_and_flush():
    gnm, prof = small(configs.cfg2, 320, 180)
    F = prepare(gnm, prof)
    F['nslots'] = 64
    ctr, atom, out4, hot, rng, pts = run_model(F, O.GEOM_4x64, 24, 8)
    total = 64 * 256 * 24
    assert int(ctr[0] + ctr[1] + ctr[2]) == total
    counts = (atom >> np.uint64(54)).astype(np.int64)
    assert int(counts.sum()) + 0 == int(ctr[0]) - 0 or int(ctr[3]) > 0
    d = F['dim']
    before = atom.copy()
    O.flush(d, atom, out4, hot)
    assert not atom.any()
    assert float(out4[:, 3].sum()) == float(ctr[0])                     # density conservation
    ysum = ((before >> np.uint64(36)) & np.uint64(0x3ffff)).astype(np.float64)
    np.testing.assert_allclose(out4[:, 0].astype(np.float64), ysum / 255.0, rtol=1e-6, atol=1e-6)
    # flush twice is idempotent on the accumulator
    again = out4.copy()
    O.flush(d, atom, out4, hot)
    assert np.array_equal(again, out4)
    # determinism: same seeds, same result
    ctr2, atom2, *_ = run_model(F, O.GEOM_4x64, 24, 8)
    assert np.array_equal(before, atom2) and np.array_equal(ctr, ctr2)


def test_hot_flags_thresholds():
    d = O.calc_dim(64, 32)
    nb = d.ah * d.astride
    atom = np.zeros(nb, np.uint64); out4 = np.zeros((nb, 4), np.float32); hot = np.zeros(nb // 16, np.uint32)
    for i, dens in enumerate([0, 128, 129, 512, 513, 2048, 2049, 1e6]):
        out4[i, 3] = dens
    O.flush(d, atom, out4, hot)
    flags = [(int(hot[i >> 4]) >> ((i & 15) * 2)) & 3 for i in range(8)]
    assert flags == [0, 0, 1, 1, 2, 2, 3, 3]
    # the flag in force weights the next flush: 1 -> x2, 2 -> x8, 3 -> x32 (iter.py:326)
    cell = ((1 << 22 | 100 << 4) << 32) | (50 << 18) | 25
    atom[:8] = np.uint64(cell) * np.uint64(3)
    before = out4[:8, 3].copy()
    O.flush(d, atom, out4, hot)
    assert (out4[:8, 3] - before).tolist() == [3 * m for m in (1, 1, 2, 2, 8, 8, 32, 32)]


@pytest.mark.parametrize('geom', ['4x64', 'ref'])
def test_device_model_matches_flam3_distribution(geom):
    """Wave-coherent selection + swap vs independent per-sample selection: same distribution."""
    g = {'4x64': O.GEOM_4x64, 'ref': O.GEOM_REF}[geom]
    gnm, prof = small(configs.cfg2, 320, 180)
    F = prepare(gnm, prof)
    F['nslots'] = 1024
    n = 1024 * 256 * 64
    ctr, atom, out4, hot, *_ = run_model(F, g, 64, 32)
    O.flush(F['dim'], atom, out4, hot)
    ref, secs, acc = O.flam3_render(F['dim'], F['packer'].prog, F['params'], F['palette'], F['seeds'], n, 8)
    d = F['dim']
    def blocks(h):
        dd = h[:, 3].reshape(d.ah, d.astride).astype(np.float64)
        H, W = d.ah // 8 * 8, d.astride // 8 * 8
        return dd[:H, :W].reshape(H // 8, 8, W // 8, 8).sum((1, 3))
    a, b = blocks(out4), blocks(ref)
    assert abs(a.sum() / n - b.sum() / n) < 3e-3
    assert np.abs(a / a.sum() - b / b.sum()).sum() < 0.03
    ca = out4[:, :3].sum(0) / out4[:, 3].sum(); cb = ref[:, :3].sum(0) / ref[:, 3].sum()
    assert np.abs(ca - cb).max() < 1.0 / 255


def test_every_variation_runs_and_is_finite_somewhere():
    """All 95 variation ids are implemented; each maps a few generic points to finite values."""
    import ctypes as C
    from cuburn_amd.genome import variations as V
    from cuburn_amd.packer import GenomePacker
    L = O.lib()
    L.ref_apply_xf.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    for name in V.var_ids:
        # non-zero affine offset: rings / fan divide by functions of it (variations.py:179-199)
        gnm = {'type': 'animation', 'xforms': {'0': {'weight': 1, 'pre_affine': {'offset': {'x': 0.3, 'y': 0.2}},
                                                     'variations': {name: dict(
                                                         [(k, dv if dv else 0.6) for k, (dv, _) in V.var_params[name].items()],
                                                         weight=0.7)}}}}
        pk = GenomePacker(gnm)
        blk = O.param_block(gnm, ['.'.join(n) for n in pk.packed], 0.5, O.calc_dim(64, 64)).astype(np.float32)
        ok = 0
        for (x0, y0) in ((0.3, 0.2), (-0.6, 0.9), (1.3, -0.4)):
            x = C.c_float(x0); y = C.c_float(y0); c = C.c_float(0.5)
            st = np.array([4294967118, 12345, 678], np.uint32)
            rc = L.ref_apply_xf(pk.prog.ctypes.data, blk.ctypes.data, 0, C.byref(x), C.byref(y), C.byref(c), st.ctypes.data)
            assert rc == 0, name
            ok += np.isfinite(x.value) and np.isfinite(y.value)
        assert ok >= 1 or name == 'pre_blur', name          # pre_blur alone outputs (0,0)
        assert abs(c.value - 0.25) < 1e-6                   # colour blend toward xform colour 0 at speed .5


def test_known_variation_values():
    import ctypes as C
    from cuburn_amd.packer import GenomePacker
    L = O.lib()
    L.ref_apply_xf.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    def apply(vars_, x0, y0, aff=None):
        xf = {'weight': 1, 'variations': vars_}
        if aff:
            xf['pre_affine'] = aff
        gnm = {'type': 'animation', 'xforms': {'0': xf}}
        pk = GenomePacker(gnm)
        blk = O.param_block(gnm, ['.'.join(n) for n in pk.packed], 0.5, O.calc_dim(64, 64)).astype(np.float32)
        x = C.c_float(x0); y = C.c_float(y0); c = C.c_float(0.0)
        st = np.array([4294967118, 12345, 678], np.uint32)
        L.ref_apply_xf(pk.prog.ctypes.data, blk.ctypes.data, 0, C.byref(x), C.byref(y), C.byref(c), st.ctypes.data)
        return x.value, y.value
    # default affine (angle 45, spread 45, magnitude 1, offset 0) is the identity up to fp32 trig error
    x, y = apply({'linear': {'weight': 1}}, 0.3, -0.2)
    assert abs(x - 0.3) < 1e-6 and abs(y + 0.2) < 1e-6
    x, y = apply({'spherical': {'weight': 1}}, 0.3, 0.4)
    assert abs(x - 0.3 / 0.25) < 1e-5 and abs(y - 0.4 / 0.25) < 1e-5
    x, y = apply({'swirl': {'weight': 1}}, 0.3, 0.4)
    s, c = np.sin(0.25), np.cos(0.25)
    assert abs(x - (s * 0.3 - c * 0.4)) < 1e-6 and abs(y - (c * 0.3 + s * 0.4)) < 1e-6
    x, y = apply({'bubble': {'weight': 2}}, 1.0, 1.0)
    assert abs(x - 2 / 1.5) < 1e-6
    # offset.y enters negated (iter.py:94)
    x, y = apply({'linear': {'weight': 1}}, 0.0, 0.0, {'offset': {'x': 0.25, 'y': 0.5}})
    assert abs(x - 0.25) < 1e-7 and abs(y + 0.5) < 1e-7


# ------------------------------------------------------------------ filters: identities
def test_filters_identities():
    d = O.calc_dim(40, 24)
    n = d.ah * d.astride
    const = np.tile(np.array([2.0, 1.0, 1.0, 4.0], np.float32), (n, 1))
    # a constant image is a fixed point of the bilateral chain and of the blurs
    out = O.bilateral_chain(d, const, 1.0, 0.05, 1.5, 0.8, 4.0)
    np.testing.assert_allclose(out, const, rtol=1e-5)
    # yuv_to_rgb of grey (u = v = 0.5 w) keeps r = g = b = y
    grey = np.tile(np.array([3.0, 2.0, 2.0, 4.0], np.float32), (n, 1))
    rgb = O.yuv_to_rgb(d, grey)
    np.testing.assert_allclose(rgb[:, :3], 3.0, rtol=1e-6)
    # logscale: zero density stays zero (NaN squashed), scale is k1*log(1+w*k2)/w
    buf = np.zeros((n, 4), np.float32); buf[5] = [1, 1, 1, 10]
    ls = O.logscale(d, buf, 2.0, 0.5)
    assert not np.isnan(ls).any() and ls[0].tolist() == [0, 0, 0, 0]
    assert abs(ls[5, 3] - 2.0 * np.log(1 + 5.0)) < 1e-5
    # colorclip: empty pixel -> zeros; alpha clamps to 1
    cc = O.colorclip(d, buf, 1.0, -1.0, 0.25, 0.01, 0.01 ** -0.75)
    assert cc[0].tolist() == [0, 0, 0, 0] and cc[5, 3] == 1.0 and cc[5, :3].max() <= 1.0
    # smearclip with nothing above 1 reduces to the plain gamma curve
    low = np.zeros((n, 4), np.float32); low[:, :] = [0.2, 0.1, 0.05, 0.5]
    sm = O.smearclip_chain(d, low, 0.7, -0.75, 0.01, 0.01 ** -0.75)
    pc = O.plainclip(d, low, -0.75, 0.01, 0.01 ** -0.75, 1.0)
    np.testing.assert_allclose(sm, pc, rtol=1e-6)


def test_blur_preserves_mass_away_from_edges():
    d = O.calc_dim(64, 48)
    n = d.ah * d.astride
    img = np.zeros((n, 4), np.float32)
    img[(d.ah // 2) * d.astride + d.astride // 2] = [1, 2, 3, 4]
    import ctypes as C
    k = O.gauss_coefs(1)
    for pattern in range(8):
        dst = np.zeros(n, np.float32)
        O.lib().ref_den_blur(C.byref(d), dst.ctypes.data, img.ctypes.data, pattern, 0, k.ctypes.data)
        assert abs(float(dst.sum()) - 4.0) < 1e-5 and (dst > 0).sum() == 7


def test_output_conversion_ranges():
    """cuburn/code/tests/test_output.py:23-53 ranges restated for the rgba formats."""
    d = O.calc_dim(32, 16)
    n = d.ah * d.astride
    rng = mwc.make_seeds(64, 9)
    for fmt, peak in ((0, 255), (1, 65535)):
        out, _ = O.f32_to_rgba(d, np.full((n, 4), -1, np.float32), rng, fmt)
        assert (out == 0).all()
        out, _ = O.f32_to_rgba(d, np.full((n, 4), 5, np.float32), rng, fmt)
        assert (out == peak).all()
        out, _ = O.f32_to_rgba(d, np.full((n, 4), 0.5, np.float32), rng, fmt)
        assert out.min() >= int(0.5 * peak) and out.max() <= int(0.5 * peak) + 1


def _plane_input(d, pixels=(), fill=0.0):
    buf = np.full((d.ah, d.astride, 4), fill, np.float32)
    for (y, x), v in pixels:
        buf[12 + y, 12 + x] = v
    return buf.reshape(-1, 4)


def test_yuv_output_known_answers():
    """The reference's own pixel-format tests (cuburn/code/tests/test_output.py:23-125) on the oracle."""
    d = O.calc_dim(640, 360)
    rng = mwc.make_seeds(4096, 11)
    # :23-53 clamping below 0 / above 1 (yuv444p): luma pinned, chroma neutral
    for fill, luma in ((-1.0, 0), (5.0, 255)):
        out, _ = O.f32_to_rgba(d, _plane_input(d, fill=fill), rng, 2)
        assert out.shape == (3, 360, 640) and out.dtype == np.uint8
        assert (out[0] == luma).all() and out[1].min() >= 127 and out[1].max() <= 128 and out[2].min() >= 127 and out[2].max() <= 128
    # :55-67 yuv444p10 zero pass-through
    out, _ = O.f32_to_rgba(d, _plane_input(d), rng, 3)
    assert (out[0] == 0).all() and (out[1] > 510).all() and (out[1] < 513).all() and (out[2] > 510).all() and (out[2] < 513).all()
    # :69-85 yuv444p10 chroma address preservation
    green = [0, 1, 0, 1]
    out, _ = O.f32_to_rgba(d, _plane_input(d, [((0, 0), green), ((1, 1), green)]), rng, 3)
    assert out[0, 0, 0] > 0 and out[0, 1, 1] > 0 and out[1, 0, 0] < 500 and out[1, 1, 1] < 500
    assert out[0, 0, 1] == 0 and 510 < out[1, 0, 1] < 513
    # :87-125 yuv420p10: chroma (0,0) from one green pixel, chroma (1,1) the mean of a green and a red one
    out, _ = O.f32_to_rgba(d, _plane_input(d, [((0, 0), green), ((2, 2), green), ((3, 3), [1, 0, 0, 1])]), rng, 4)
    w, h = 640, 360
    assert out.shape == (w * h * 6 // 4,)
    luma = out[:w * h].reshape(h, w)
    cb = out[w * h:w * h + w * h // 4].reshape(h // 2, w // 2)
    cr = out[w * h + w * h // 4:].reshape(h // 2, w // 2)
    assert luma[0, 0] > 0 and luma[1, 0] == 0 and luma[0, 1] == 0 and luma[1, 1] == 0 and luma[2, 2] > 0 and luma[3, 3] > 0
    assert 172 <= cb[0, 0] <= 174 and 511 <= cb[0, 1] <= 512 and 511 <= cb[1, 0] <= 512
    # green + red, equal alpha: cb = (0.168736 + 0.331264) / 2 ... evaluated by hand
    want_cb = 1023 * ((-0.331264 - 0.168736) / 2 + 0.5)
    want_cr = 1023 * ((-0.418688 + 0.5) / 2 + 0.5)
    assert want_cb <= cb[1, 1] <= want_cb + 1 and want_cr - 1 <= cr[1, 1] <= want_cr + 1
    # 12-bit studio swing: black = 256 / 2048 / 2048, white = 3760 / 2048 / 2048 (+ dither below one code)
    out, _ = O.f32_to_rgba(d, _plane_input(d, fill=0.0), rng, 5)
    assert (out[0] == 256).all() and (np.abs(out[1].astype(int) - 2048) <= 1).all() and (np.abs(out[2].astype(int) - 2048) <= 1).all()
    out, _ = O.f32_to_rgba(d, _plane_input(d, fill=7.0), rng, 5)
    assert (out[0] == 3760).all() and (np.abs(out[1].astype(int) - 2048) <= 1).all()


def test_yuv_output_rng_use():
    """Three draws per pixel (Y, Cb, Cr) wherever the value is positive; in 4:2:0 only the top-left
    quadrant's pixels draw for chroma.  States advance exactly that often."""
    d = O.calc_dim(16, 8)
    n = d.ah * d.astride
    rng = mwc.make_seeds(32, 3)
    buf = np.full((n, 4), 0.5, np.float32)
    for fmt, draws in ((2, 3 * 128), (3, 3 * 128), (5, 3 * 128), (4, 128 + 2 * 32)):
        _, after = O.f32_to_rgba(d, buf, rng, fmt)
        steps = 0
        for t in range(32):
            mul, state, carry = (int(v) for v in rng[t])
            k = 0
            while (state, carry) != (int(after[t, 1]), int(after[t, 2])):
                v = mul * state + carry                  # cuburn/code/mwc.py:56-63
                state, carry = v & 0xffffffff, v >> 32
                k += 1
                assert k <= 64
            steps += k
        assert steps == draws, (fmt, steps, draws)
