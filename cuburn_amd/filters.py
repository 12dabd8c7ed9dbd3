"""
Host side of the filter chain: registry, per-filter scalar derivation, launch order.

Same classes and ``apply`` contract as cuburn/filters.py:23-198 — every filter leaves its
result in the front buffer — but a filter here is one call into libflame_hip (``fl_filter``),
which owns the kernels and scratch buffers.
"""
import numpy as np
from numpy import float32 as f32

from . import _lib


class Filter(object):
    filter_map = {}
    name = ''
    full_side = False

    def apply(self, fb, gprof, params, dim, tc, stream=None):
        raise NotImplementedError()

    def scalars(self, gprof, params, dim, tc):
        """The kernel arguments derived on the host, as a float32 list."""
        return []

    def _run(self, fb, dim, vals):
        arr = np.asarray(vals, dtype=np.float32)
        _lib.check(_lib.load().fl_filter(fb.ctx, _lib.FILT[self.name], dim.w, dim.h,
                                         arr.ctypes.data, len(arr)))

    @classmethod
    def register(cls, name):
        def register_(subcls):
            cls.filter_map[name] = subcls
            subcls.name = name
            return subcls
        return register_


class _Simple(Filter):
    def apply(self, fb, gprof, params, dim, tc, stream=None):
        self._run(fb, dim, self.scalars(gprof, params, dim, tc))


@Filter.register('yuv')
class YuvFilterLib(_Simple):
    pass


@Filter.register('bilateral')
class Bilateral(_Simple):
    radius = 15
    directions = 8

    def scalars(self, gprof, params, dim, tc):
        # spatial parameter scaled so a "pixel" is a 1080p pixel (cuburn/filters.py:74-76)
        sstd = params.spatial_std(tc) * dim.w / 1920.
        return [f32(sstd), f32(params.color_std(tc)), f32(params.density_std(tc)),
                f32(params.density_pow(tc)), f32(params.gradient(tc))]


@Filter.register('logscale')
class Logscale(_Simple):
    def scalars(self, gprof, params, dim, tc):
        k1 = f32(params.brightness(tc) * 268 / 256)
        area = dim.h / (params.scale(tc) ** 2 * dim.w)       # cuburn/filters.py:103-106
        k2 = f32(1.0 / (area * gprof.spp(tc)))
        return [k1, k2]


def calc_lingam(params, tc):
    """gamma / linear-range scalars shared by the clip family (cuburn/filters.py:132-136)."""
    gam = f32(1 / params.gamma(tc))
    lin = f32(params.gamma_threshold(tc))
    lingam = f32(lin ** (gam - 1.0) if lin > 0 else 0)
    return gam, lin, lingam


@Filter.register('haloclip')
class HaloClip(_Simple):
    def scalars(self, gprof, params, dim, tc):
        return [f32(1 / gprof.filters.colorclip.gamma(tc) - 1)]


@Filter.register('smearclip')
class SmearClip(_Simple):
    full_side = True

    def scalars(self, gprof, params, dim, tc):
        gam, lin, lingam = calc_lingam(gprof.filters.colorclip, tc)
        return [f32(params.width(tc)), f32(gam - 1), lin, lingam]


@Filter.register('colorclip')
class ColorClip(_Simple):
    def scalars(self, gprof, params, dim, tc):
        gam, lin, lingam = calc_lingam(params, tc)
        return [f32(params.vibrance(tc)), f32(params.highlight_power(tc)), gam, lin, lingam]


@Filter.register('plainclip')
class PlainClip(_Simple):
    def scalars(self, gprof, params, dim, tc):
        gam, lin, lingam = calc_lingam(gprof.filters.colorclip, tc)
        return [f32(gam - 1), lin, lingam, f32(gprof.filters.plainclip.brightness(tc))]


@Filter.register('logencode')
class LogEncode(_Simple):
    def scalars(self, gprof, params, dim, tc):
        return [f32(params.degamma(tc))]


def create(gprof):
    order = ['yuv'] + list(gprof.filter_order)
    return [Filter.filter_map[f]() for f in order]
