"""
oracle.py — Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker / reported baseline (oracle/flame_ref.h).  Nothing under cuburn_amd/
imports it.

Two parts:
  * ctypes wrappers of libflame_ref.so (flame_ref.c, filters_ref.c): RNG, device-model
    iterate / flush, flam3-style baseline, every filter, output conversion;
  * an independent numpy restatement of the per-temporal-sample parameter preparation
    (cuburn/code/interp.py:234-272 + precalc of cuburn/code/iter.py:12-30,56-95 and
    cuburn/code/variations.py), evaluated BY NAME from the genome so that it does not depend
    on the product's layout code: `param_block(gnm, names, t, dim)`.

Schema defaults come from the reference-generated fixtures tests/golden/spec_defaults.json and
tests/golden/var_spec.json.
"""
import ctypes as C
import json
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
LIB_PATH = os.path.join(HERE, 'libflame_ref.so')


def build():
    subprocess.run(['make', '-C', HERE, '-s'], check=True)


class ref_dim(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ('w', 'h', 'aw', 'ah', 'astride')]


class ref_geom(C.Structure):
    _fields_ = [('nw', C.c_int), ('wl', C.c_int), ('ref_shuffle', C.c_int)]


GEOM_REF = ref_geom(8, 32, 1)       # the reference's 8 warps x 32 lanes (iter.py:106,275-278)
GEOM_4x64 = ref_geom(4, 64, 0)      # MI355X kernel: 4 waves x 64 lanes
GEOM_8x64 = ref_geom(8, 64, 0)
GEOM_16x64 = ref_geom(16, 64, 0)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.ref_mwc_next_01.restype = C.c_float
        L.ref_mwc_next_11.restype = C.c_float
        L.ref_mwc_next.restype = C.c_uint32
        L.ref_catmull_rom.restype = C.c_float
        L.ref_catmull_rom.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int]
        L.ref_flam3_render.restype = C.c_double
        L.ref_flam3_render.argtypes = [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 2 + [C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_iter_launch.restype = C.c_int
        L.ref_iter_launch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 3 + [C.c_uint32] * 3 + [C.c_void_p]
        L.ref_interp_palette.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        for n in ('ref_den_blur', 'ref_den_blur_1c', 'ref_full_blur'):
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ref_bilateral.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_float] * 5
        L.ref_bilateral_chain.argtypes = [C.c_void_p] * 5 + [C.c_float] * 5
        L.ref_logscale.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float]
        L.ref_colorclip.argtypes = [C.c_void_p, C.c_void_p] + [C.c_float] * 5
        L.ref_smearclip_chain.argtypes = [C.c_void_p] * 5 + [C.c_float] * 3
        L.ref_haloclip_chain.argtypes = [C.c_void_p] * 5 + [C.c_float]
        L.ref_plainclip.argtypes = [C.c_void_p, C.c_void_p] + [C.c_float] * 4
        L.ref_logencode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.ref_yuv_to_rgb.argtypes = [C.c_void_p] * 3
        L.ref_f32_to_rgba.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.ref_flush.argtypes = [C.c_void_p] * 4
        L.ref_shuffle_perm.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.ref_mwc_sums.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.ref_mwc_stream.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.ref_calc_dim.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.ref_unpack_cell.argtypes = [C.c_uint64, C.c_void_p]
        L.ref_var_supported.argtypes = [C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ small wrappers
def calc_dim(w, h):
    d = ref_dim()
    lib().ref_calc_dim(w, h, C.byref(d))
    return d


def mwc_sums(seeds, rounds):
    s = np.ascontiguousarray(seeds, dtype=np.uint32).copy()
    out = np.zeros(len(s), dtype=np.uint64)
    lib().ref_mwc_sums(_p(s), len(s), rounds, _p(out))
    return out, s


def mwc_stream(seed_row, n):
    s = np.ascontiguousarray(seed_row, dtype=np.uint32).copy()
    out = np.zeros(n, dtype=np.uint32)
    lib().ref_mwc_stream(_p(s), n, _p(out))
    return out


def shuffle_perm(geom, rnd):
    out = np.zeros(geom.nw * geom.wl, dtype=np.uint32)
    lib().ref_shuffle_perm(C.byref(geom), rnd, _p(out))
    return out


def unpack_cell(cell):
    out = np.zeros(4, dtype=np.uint32)
    lib().ref_unpack_cell(C.c_uint64(int(cell)), _p(out))
    return out           # [sumY, sumU, sumV, count]


def catmull_rom(times, knots, t, mag=False):
    tt = np.full(32, 1e9, dtype=np.float32)
    kk = np.zeros(32, dtype=np.float32)
    tt[:len(times)] = times
    kk[:len(knots)] = knots
    return float(lib().ref_catmull_rom(_p(tt), _p(kk), C.c_float(t), int(mag)))


def interp_palette(pal_rgba, pal_times, ts, td, rng64x256):
    pal = np.zeros((32, 256, 4), dtype=np.float32)
    pal[:len(pal_rgba)] = pal_rgba
    pt = np.full(32, 1e9, dtype=np.float32)
    pt[:len(pal_times)] = pal_times
    rng = np.ascontiguousarray(rng64x256, dtype=np.uint32).copy()
    out = np.zeros((64, 256), dtype=np.uint64)
    lib().ref_interp_palette(_p(pal), _p(pt), len(pal_rgba), ts, td, _p(rng), _p(out))
    return out, rng


def iter_launch(geom, dim, prog, params, palette, rng, points, nslots, hot, atom, out4,
                round0, nrounds, fuse):
    """In-place on rng, points, atom, out4; returns counters [accepted, oob, dropped, spills].
    ``params`` holds one block per slot (temporal sample s = slot s)."""
    assert params.shape[0] == nslots, 'one parameter block per slot'
    params = np.ascontiguousarray(params, dtype=np.float32)
    ctr = np.zeros(4, dtype=np.uint64)
    prog = np.ascontiguousarray(prog, dtype=np.int32)
    rc = lib().ref_iter_launch(C.byref(geom), C.byref(dim), _p(prog), _p(params), _p(palette), _p(rng),
                               _p(points), nslots, _p(hot), _p(atom), _p(out4), round0, nrounds, fuse, _p(ctr))
    if rc:
        raise ValueError('oracle: unsupported variation in program')
    return ctr


def flush(dim, atom, out4, hot):
    lib().ref_flush(C.byref(dim), _p(atom), _p(out4), _p(hot))


def flam3_render(dim, prog, params, palette, seeds, nsamples, nthreads, fuse=15):
    """flam3-style CPU chaos game; returns (float4 histogram, seconds, accepted).
    ``params`` holds one block per temporal sample (any count); all get equal weight."""
    params = np.ascontiguousarray(params, dtype=np.float32)
    assert params.ndim == 2
    nbins = dim.ah * dim.astride
    out = np.zeros((nbins, 4), dtype=np.float32)
    acc = C.c_uint64()
    prog = np.ascontiguousarray(prog, dtype=np.int32)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    secs = lib().ref_flam3_render(C.byref(dim), _p(prog), _p(params), params.shape[0], _p(palette), _p(seeds), len(seeds),
                                  int(nsamples), int(nthreads), int(fuse), _p(out), C.byref(acc))
    return out, secs, acc.value


# ------------------------------------------------------------------ filters (arrays are float32, modified in place)
def yuv_to_rgb(dim, src):
    dst = np.empty_like(src)
    lib().ref_yuv_to_rgb(C.byref(dim), _p(dst), _p(src))
    return dst


def gauss_coefs(stdev):
    """cuburn/filters.py:11-16"""
    c = np.exp(np.float32(np.arange(-3, 4)) ** 2 / np.float32(-2 * stdev ** 2)).astype(np.float32)
    return (c / np.sum(c)).astype(np.float32)


def bilateral_chain(dim, front, sstd, cstd, dstd, dpow, gspeed):
    front = np.ascontiguousarray(front, dtype=np.float32).copy()
    back = np.zeros_like(front)
    side = np.zeros(front.size // 4 * 4, dtype=np.float32)
    k = gauss_coefs(1)
    lib().ref_bilateral_chain(C.byref(dim), _p(front), _p(back), _p(side), _p(k), sstd, cstd, dstd, dpow, gspeed)
    return front


def bilateral_pass(dim, src4, pattern, sstd, cstd, dstd, dpow, gspeed, radius=15):
    """One direction: den_blur -> den_blur_1c -> bilateral; returns (dst4, blur1, blur2)."""
    n = dim.ah * dim.astride
    k = gauss_coefs(1)
    b1 = np.zeros(n, np.float32); b2 = np.zeros(n, np.float32); dst = np.zeros((n, 4), np.float32)
    L = lib()
    L.ref_den_blur(C.byref(dim), _p(b1), _p(src4), pattern, 0, _p(k))
    L.ref_den_blur_1c(C.byref(dim), _p(b2), _p(b1), pattern, 1, _p(k))
    L.ref_bilateral(C.byref(dim), _p(dst), _p(src4), _p(b2), pattern, radius, sstd, cstd, dstd, dpow, gspeed)
    return dst, b1, b2


def logscale(dim, buf, k1, k2):
    buf = buf.copy(); lib().ref_logscale(C.byref(dim), _p(buf), k1, k2); return buf


def colorclip(dim, buf, vib, highpow, gam, lin, lingam):
    buf = buf.copy(); lib().ref_colorclip(C.byref(dim), _p(buf), vib, highpow, gam, lin, lingam); return buf


def smearclip_chain(dim, front, width, gam_m_1, lin, lingam):
    front = front.copy(); back = np.zeros_like(front); side = np.zeros_like(front)
    k = gauss_coefs(width)
    lib().ref_smearclip_chain(C.byref(dim), _p(front), _p(back), _p(side), _p(k), gam_m_1, lin, lingam)
    return front


def haloclip_chain(dim, front, gam_m_1):
    front = front.copy(); back = np.zeros_like(front); side = np.zeros_like(front)
    k = gauss_coefs(1)
    lib().ref_haloclip_chain(C.byref(dim), _p(front), _p(back), _p(side), _p(k), gam_m_1)
    return front


def plainclip(dim, buf, gam_m_1, lin, lingam, brightness):
    buf = buf.copy(); lib().ref_plainclip(C.byref(dim), _p(buf), gam_m_1, lin, lingam, brightness); return buf


def logencode(dim, src, degamma):
    dst = np.empty_like(src); lib().ref_logencode(C.byref(dim), _p(dst), _p(src), degamma); return dst


def f32_to_rgba(dim, src, rng, fmt):
    rng = np.ascontiguousarray(rng, dtype=np.uint32).copy()
    if fmt < 2:
        dst = np.zeros((dim.h, dim.w, 4), dtype=np.uint16 if fmt else np.uint8)
    elif fmt == 4:
        dst = np.zeros(dim.h * dim.w * 6 // 4, dtype=np.uint16)
    else:
        dst = np.zeros((3, dim.h, dim.w), dtype=np.uint8 if fmt == 2 else np.uint16)
    lib().ref_f32_to_rgba(C.byref(dim), _p(src), _p(rng), len(rng), fmt, _p(dst))
    return dst, rng


# ------------------------------------------------------------------ parameter preparation by name
_spec_defaults = None
_var_spec = None


def _defaults():
    global _spec_defaults, _var_spec
    if _spec_defaults is None:
        _spec_defaults = json.load(open(os.path.join(GOLDEN, 'spec_defaults.json')))
        _var_spec = json.load(open(os.path.join(GOLDEN, 'var_spec.json')))
    return _spec_defaults, _var_spec


def normalize(knots, scale):
    """cuburn/genome/use.py:129-158, restated."""
    if isinstance(knots, (int, float)):
        v0 = v1 = 0.0
        pts = [(0.0, float(knots)), (1.0, float(knots))]
    elif len(knots) == 2:
        v0 = v1 = 0.0
        pts = [(0.0, knots[0]), (1.0, knots[1])]
    else:
        assert len(knots) % 2 == 0
        p0, v0, p1, v1 = knots[:4]
        pts = [(0.0, p0), (1.0, p1)] + [(knots[i], knots[i + 1]) for i in range(4, len(knots), 2)]
    v0 *= scale
    v1 *= scale
    pts.sort()
    if pts[0][0] >= 0:
        pts.insert(0, (-2.0, pts[1][1] - (pts[1][0] + 2.0) * v0))
    if pts[-1][0] <= 1:
        pts.append((3.0, pts[-2][1] + (3.0 - pts[-2][0]) * v1))
    return [p[0] for p in pts], [p[1] for p in pts]


class _Genome(object):
    """Evaluate genome splines by path at time t (device semantics: float32 Catmull-Rom)."""

    def __init__(self, gnm, t):
        self.gnm, self.t = gnm, t
        self.scale = gnm.get('time', {}).get('duration', 1)

    def val(self, path, default, interp):
        node = self.gnm
        for k in path:
            if not isinstance(node, dict) or k not in node:
                node = default
                break
            node = node[k]
        times, knots = normalize(node, self.scale)
        return catmull_rom(times, knots, self.t, interp == 'mag')


def _affine_vals(G, base, sd):
    """cuburn/code/iter.py:81-95"""
    def v(sub):
        d, i = sd['xform.pre_affine.' + '.'.join(sub)]
        return G.val(base + sub, d, i)
    pri = np.float32(v(('angle',))) * np.float32(math.pi) / np.float32(180.0)
    spr = np.float32(v(('spread',))) * np.float32(math.pi) / np.float32(180.0)
    magx, magy = v(('magnitude', 'x')), v(('magnitude', 'y'))
    return {'xx': magx * math.cos(pri - spr), 'yx': -magx * math.sin(pri - spr),
            'xy': -magy * math.cos(pri + spr), 'yy': magy * math.sin(pri + spr),
            'xo': v(('offset', 'x')), 'yo': -v(('offset', 'y'))}


def param_block(gnm, names, t, dim):
    """
    Values of the parameter-block floats named by ``names`` (dotted paths as in the packer's
    ``packed`` list) at time ``t``; float64 formulas over float32 spline evaluations.
    """
    sd, vs = _defaults()
    G = _Genome(gnm, t)
    cache = {}
    keys = sorted(gnm['xforms'].keys())
    out = []
    def bits(i):
        return float(np.array([i], dtype=np.int32).view(np.float32)[0])
    for name in names:
        p = tuple(name.split('.')) if isinstance(name, str) else tuple(name)
        if p[0] == 'pad':
            out.append(0.0)
            continue
        if p[-1] in ('#nvar', '#id'):              # integer structure words, include/flame_hip.h (5)
            base = gnm['xforms'][p[1]] if p[0] == 'xforms' else gnm['final_xform']
            if p[-1] == '#nvar':
                out.append(bits(len(base.get('variations', {})) | ((1 if 'post_affine' in base else 0) << 8)))
            else:
                out.append(bits(vs[p[-2]]['num']))
            continue
        if p[0] == 'camera':                       # cuburn/code/iter.py:56-79
            if 'cam' not in cache:
                rot = np.float32(G.val(('camera', 'rotation'), *sd['camera.rotation'])) * np.float32(math.pi) / np.float32(180.0)
                rs, rc = math.sin(rot), math.cos(rot)
                cx = G.val(('camera', 'center', 'x'), *sd['camera.center.x'])
                cy = G.val(('camera', 'center', 'y'), *sd['camera.center.y'])
                s = G.val(('camera', 'scale'), *sd['camera.scale']) * dim.w
                cache['cam'] = {'xx': s * rc, 'xy': -s * rs, 'xo': s * (rs * cy - rc * cx) + 0.5 * dim.aw,
                                'yx': s * rs, 'yy': s * rc, 'yo': -s * (rs * cx + rc * cy) + 0.5 * dim.ah}
            out.append(cache['cam'][p[1]])
        elif p[0] == 'den':                        # cuburn/code/iter.py:12-30
            if 'den' not in cache:
                w = [G.val(('xforms', k, 'weight'), *sd['xform.weight']) for k in keys]
                tot = sum(w)
                acc, cdf = 0.0, {}
                for k, wk in zip(keys, w):
                    acc += wk / tot
                    cdf[k] = acc
                cache['den'] = cdf
            out.append(cache['den'][p[1]])
        else:
            base = ('xforms', p[1]) if p[0] == 'xforms' else ('final_xform',)
            rest = p[2:] if p[0] == 'xforms' else p[1:]
            if rest[0] in ('pre_affine', 'post_affine'):
                ck = base + (rest[0],)
                if ck not in cache:
                    cache[ck] = _affine_vals(G, ck, sd)
                out.append(cache[ck][rest[1]])
            elif rest[0] in ('color', 'color_speed'):
                out.append(G.val(base + rest, *sd['xform.' + rest[0]]))
            else:
                assert rest[0] == 'variations'
                vname, pname = rest[1], rest[2]
                spec = vs[vname]['params']
                vb = base + ('variations', vname)
                def gv(n):
                    return G.val(vb + (n,), *spec[n])
                if pname in spec:
                    out.append(gv(pname))
                elif pname == 'cn':                                  # variations.py:292-294
                    out.append(gv('dist') / (2.0 * gv('power')))
                elif pname in ('dx2', 'dy2'):                        # variations.py:136-140
                    o = G.val(base + ('pre_affine', 'offset', pname[1]), *sd['xform.pre_affine.offset.' + pname[1]])
                    out.append(1.0 / (o * o + 1.0e-20))
                elif pname in ('mdist', 'sin', 'cos'):               # variations.py:267-273
                    pang = gv('angle') * (math.pi / 2)
                    pd = max(1e-9, gv('dist'))
                    out.append({'mdist': pd, 'sin': math.sin(pang), 'cos': pd * math.cos(pang)}[pname])
                elif pname in ('x2', 'y2'):                          # variations.py:630-634
                    l = gv(pname[0] + 'length')
                    out.append(1.0 / max(1e-20, l * l))
                else:
                    raise KeyError(name)
    return np.array(out, dtype=np.float64)      # structure words carry int32 bit patterns (exact in f64)
