"""
The synthetic BASELINE.json configurations as cuburn animation genomes + profiles.

Deterministic restatements of the five ``configs`` of BASELINE.json:
genome dicts in cuburn's JSON animation format and matching profile dicts.  Used by
bench.py, __graft_entry__.smoke() and the parity tests.
"""
import numpy as np
from .genome.util import palette_encode


def _affine(angle=0.0, scale=1.0, ox=0.0, oy=0.0, spread=45.0, sx=None, sy=None):
    # angle/spread semantics of cuburn/code/iter.py:81-95: angle 45 + spread 45 = identity axes
    return {'angle': 45.0 + angle, 'spread': spread,
            'magnitude': {'x': scale if sx is None else sx, 'y': scale if sy is None else sy},
            'offset': {'x': ox, 'y': oy}}


def grey_ramp():
    t = np.linspace(0, 1, 256)
    return np.stack([t, t, t], 1)


def fire_palette():
    t = np.linspace(0, 1, 256)
    return np.stack([np.clip(1.6 * t, 0, 1), np.clip(1.9 * t - 0.55, 0, 1) ** 1.2, np.clip(3.0 * t - 2.0, 0, 1)], 1)


def ice_palette():
    t = np.linspace(0, 1, 256)
    return np.stack([np.clip(2.5 * t - 1.4, 0, 1), np.clip(1.5 * t - 0.2, 0, 1), np.clip(0.25 + 1.1 * t, 0, 1)], 1)


def _pal(time, data):
    return [float(time)] + palette_encode(data)


def cfg1():
    """512x512 still, 2-xform linear-only flame, 1M samples (CPU plumbing case)."""
    gnm = {
        'type': 'animation', 'name': 'cfg1-linear2',
        'camera': {'center': {'x': 0.0, 'y': 0.0}, 'rotation': 0.0, 'scale': 0.25},
        'time': {'duration': 1, 'frame_width': 0.0},
        'palette': [_pal(0.0, grey_ramp())],
        'xforms': {
            '0': {'weight': 0.5, 'color': 0.0, 'color_speed': 0.5,
                  'pre_affine': _affine(0, 0.5, -0.5, 0.0), 'variations': {'linear': {'weight': 1.0}}},
            '1': {'weight': 0.5, 'color': 1.0, 'color_speed': 0.5,
                  'pre_affine': _affine(0, 0.5, 0.5, 0.0), 'variations': {'linear': {'weight': 1.0}}},
        },
    }
    prof = {'width': 512, 'height': 512, 'spp': 1.0e6 / (512 * 512), 'fps': 1, 'duration': 1,
            'frame_width': 0, 'start': None, 'end': None, 'output': {'type': 'raw'},
            'filter_order': ['bilateral', 'logscale', 'colorclip']}
    return gnm, prof


def _cfg2_xforms():
    return {
        '0': {'weight': 0.5, 'color': 0.0, 'color_speed': 0.5,
              'pre_affine': _affine(25.0, 0.62, 0.45, 0.15), 'variations': {'linear': {'weight': 1.0}}},
        '1': {'weight': 0.3, 'color': 0.55, 'color_speed': 0.5,
              'pre_affine': _affine(-35.0, 0.85, -0.55, 0.35), 'variations': {'spherical': {'weight': 1.0}}},
        '2': {'weight': 0.2, 'color': 1.0, 'color_speed': 0.5,
              'pre_affine': _affine(70.0, 0.75, 0.1, -0.6),
              'variations': {'swirl': {'weight': 0.5}, 'linear': {'weight': 0.5}}},
    }


def cfg2(samples=2 ** 28):
    """1920x1080 still, 3-xform (linear + spherical + swirl), 256M samples — the headline config."""
    gnm = {
        'type': 'animation', 'name': 'cfg2-lin-sph-swirl',
        'camera': {'center': {'x': 0.0, 'y': 0.0}, 'rotation': 0.0, 'scale': 0.25},
        'time': {'duration': 1, 'frame_width': 0.0},
        'palette': [_pal(0.0, grey_ramp())],
        'xforms': _cfg2_xforms(),
    }
    prof = {'width': 1920, 'height': 1080, 'spp': samples / (1920.0 * 1080.0), 'fps': 1, 'duration': 1,
            'frame_width': 0, 'output': {'type': 'raw'},
            'filter_order': ['bilateral', 'logscale', 'colorclip']}
    return gnm, prof


def cfg3(samples=2 ** 30):
    """1920x1080, 8 xforms + final xform, two palettes (interpolated), DE filter on, 1B samples."""
    rs = np.random.RandomState(3)
    names = ['linear', 'sinusoidal', 'spherical', 'swirl', 'horseshoe', 'polar', 'bubble', 'eyefish']
    xforms = {}
    for i, v in enumerate(names):
        ang = float(rs.uniform(-180, 180))
        xforms[str(i)] = {
            'weight': float(rs.uniform(0.5, 1.5)), 'color': i / 7.0, 'color_speed': 0.5,
            'pre_affine': _affine(ang, float(rs.uniform(0.45, 0.8)), float(rs.uniform(-0.7, 0.7)), float(rs.uniform(-0.5, 0.5))),
            'variations': {v: {'weight': 0.7}, 'linear': {'weight': 0.3}} if v != 'linear' else {'linear': {'weight': 1.0}},
        }
    # a rotating pre affine on one xform so temporal samples differ (animation format: [p0, v0, p1, v1])
    xforms['3']['pre_affine']['angle'] = [60.0, 40.0, 100.0, 40.0]
    gnm = {
        'type': 'animation', 'name': 'cfg3-8xf-final',
        'camera': {'center': {'x': 0.0, 'y': 0.0}, 'rotation': [0.0, 10.0, 10.0, 10.0], 'scale': 0.22},
        'time': {'duration': 1, 'frame_width': 1.0},
        'palette': [_pal(0.0, fire_palette()), _pal(1.0, ice_palette())],
        'xforms': xforms,
        'final_xform': {'color': 0.0, 'color_speed': 0.0, 'pre_affine': _affine(5.0, 1.02, 0.0, 0.0),
                        'variations': {'linear': {'weight': 1.0}}},
    }
    prof = {'width': 1920, 'height': 1080, 'spp': samples / (1920.0 * 1080.0), 'fps': 24, 'duration': 2,
            'frame_width': 1.0, 'output': {'type': 'raw'},
            'filter_order': ['bilateral', 'logscale', 'colorclip']}
    return gnm, prof


def cfg4(samples=2 ** 28):
    """3840x2160 animation, 60 frames temporally sampled; frames shard across GPUs."""
    gnm, _ = cfg3()
    gnm = dict(gnm, name='cfg4-4k-anim')
    prof = {'width': 3840, 'height': 2160, 'spp': samples / (3840.0 * 2160.0), 'fps': 30, 'duration': 2,
            'frame_width': 1.0, 'output': {'type': 'raw'},
            'filter_order': ['bilateral', 'logscale', 'colorclip']}
    return gnm, prof


def cfg5(samples=2 ** 32):
    """7680x4320, 12-xform heavy-variation flame, 4B samples/frame (HBM / contention stress)."""
    rs = np.random.RandomState(5)
    heavy = [('julian', {'power': 3.0, 'dist': 1.0}), ('juliascope', {'power': 2.0, 'dist': 1.0}),
             ('ngon', {'sides': 5.0, 'power': 3.0, 'circle': 1.0, 'corners': 2.0}),
             ('super_shape', {'m': 4.0, 'n1': 1.0, 'n2': 1.0, 'n3': 1.0, 'rnd': 0.1, 'holes': 0.0}),
             ('cpow', {'r': 1.0, 'i': 0.1, 'power': 2.0}), ('escher', {'beta': 0.4}), ('elliptic', {}),
             ('bipolar', {'shift': 0.2}), ('wedge', {'angle': 0.4, 'hole': 0.1, 'count': 3.0, 'swirl': 0.2}),
             ('flux', {'spread': 0.3}),
             ('mobius', {'re_a': 0.8, 'im_a': 0.1, 're_b': 0.2, 'im_b': 0.0, 're_c': 0.1, 'im_c': 0.2, 're_d': 1.0, 'im_d': 0.0}),
             ('gaussian_blur', {})]
    xforms = {}
    for i, (v, params) in enumerate(heavy):
        vd = dict(params, weight=0.6)
        xforms['%02d' % i] = {
            'weight': float(rs.uniform(0.5, 1.5)), 'color': i / 11.0, 'color_speed': 0.4,
            'pre_affine': _affine(float(rs.uniform(-180, 180)), float(rs.uniform(0.5, 0.85)),
                                  float(rs.uniform(-0.6, 0.6)), float(rs.uniform(-0.4, 0.4))),
            'variations': {v: vd, 'linear': {'weight': 0.4}},
        }
    gnm = {
        'type': 'animation', 'name': 'cfg5-12xf-heavy',
        'camera': {'center': {'x': 0.0, 'y': 0.0}, 'rotation': 0.0, 'scale': 0.2},
        'time': {'duration': 1, 'frame_width': 0.0},
        'palette': [_pal(0.0, fire_palette())],
        'xforms': xforms,
    }
    prof = {'width': 7680, 'height': 4320, 'spp': samples / (7680.0 * 4320.0), 'fps': 1, 'duration': 1,
            'frame_width': 0, 'output': {'type': 'raw'},
            'filter_order': ['bilateral', 'logscale', 'colorclip']}
    return gnm, prof


def allvars():
    """Not a BASELINE config: a genome that uses EVERY variation — twelve xforms of eight variations each in table order, every
    parameter of every parametric variation away from its default (some animated), post affines on every other xform, a final
    xform with its own post affine — at a small size.  The packer's knot rows for it are pinned to the reference's
    GenomePacker.pack (tests/golden/packer.json, tests/golden/make_golden.py section 4: cuburn/code/interp.py:125-232,
    cuburn/genome/variations.py:7-16); the config genomes alone reach 15 of the 95 variations' parameter rows."""
    from .genome import variations as V
    names = [name for _, name, _ in V._TABLE]
    rs = np.random.RandomState(95)
    xforms = {}
    for i in range(0, len(names), 8):
        vs = {}
        for j, v in enumerate(names[i:i + 8]):
            vd = {'weight': round(0.05 + 0.02 * j, 4)}
            for k, (pn, (default, interp)) in enumerate(sorted((pn, pd) for pn, pd in V.var_params[v].items() if pn != 'weight')):
                val = round(float(default) + 0.13 * (k + 1) + 0.01 * j, 4)
                # every third parameter moves during the frame (animation format [p0, v0, p1, v1]; mag-domain ones stay positive)
                vd[pn] = [val, 0.1, round(val + 0.25, 4), 0.1] if (k + j) % 3 == 0 else val
            vs[v] = vd
        key = '%02d' % (i // 8)
        xf = {'weight': float(rs.uniform(0.5, 1.5)), 'color': (i // 8) / 11.0, 'color_speed': 0.3 + 0.02 * (i // 8),
              'pre_affine': _affine(float(rs.uniform(-180, 180)), float(rs.uniform(0.5, 0.85)),
                                    float(rs.uniform(-0.6, 0.6)), float(rs.uniform(-0.4, 0.4))),
              'variations': vs}
        if (i // 8) % 2 == 1:
            xf['post_affine'] = _affine(float(rs.uniform(-30, 30)), float(rs.uniform(0.9, 1.1)), float(rs.uniform(-0.1, 0.1)), float(rs.uniform(-0.1, 0.1)))
        xforms[key] = xf
    xforms['03']['pre_affine']['angle'] = [20.0, 30.0, 50.0, 30.0]
    gnm = {
        'type': 'animation', 'name': 'allvars-95',
        'camera': {'center': {'x': 0.05, 'y': -0.1}, 'rotation': [5.0, 8.0, 13.0, 8.0], 'scale': 0.21},
        'time': {'duration': 1, 'frame_width': 1.0},
        'palette': [_pal(0.0, fire_palette()), _pal(1.0, ice_palette())],
        'xforms': xforms,
        'final_xform': {'color': 0.5, 'color_speed': 0.1, 'pre_affine': _affine(3.0, 1.01, 0.02, -0.01),
                        'post_affine': _affine(-2.0, 0.99, 0.0, 0.01),
                        'variations': {'linear': {'weight': 0.9}, 'curl': {'weight': 0.1, 'c1': 0.3, 'c2': [0.1, 0.0, 0.2, 0.0]}}},
    }
    prof = {'width': 640, 'height': 360, 'spp': 2 ** 22 / (640.0 * 360.0), 'fps': 24, 'duration': 2,
            'frame_width': 1.0, 'output': {'type': 'raw'},
            'filter_order': ['bilateral', 'logscale', 'colorclip']}
    return gnm, prof


CONFIGS = {'cfg1': cfg1, 'cfg2': cfg2, 'cfg3': cfg3, 'cfg4': cfg4, 'cfg5': cfg5}
