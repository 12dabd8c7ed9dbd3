"""
Document schemas: animation / node / edge genomes and render profiles.

Restates cuburn/genome/specs.py:4-139 (keys, defaults, interpolation domains) for the
documents that cross the render boundary.  Authoring-only fields are kept so that real
cuburn JSON files validate, but nothing here blends or converts genomes.
"""
from .spectypes import (spline, scalespline, scalar, refscalar, map_, list_, enum,
                        String, Palette, XYPair, Scalar, RefScalar)
from . import variations as _v

_var_specs = dict(
    (name, dict((k, (scalespline(d, var=(k != 'weight')) if i == 'mag'
                     else spline(d, var=(k != 'weight'))))
                for k, (d, i) in params.items()))
    for name, params in _v.var_params.items())

affine = {
    'angle': spline(45, period=360),
    'spread': spline(45, period=360),
    'magnitude': XYPair(scalespline()),
    'offset': XYPair(spline()),
}

xform = {
    'pre_affine': affine,
    'post_affine': affine,
    'color': spline(0, 0, 1),
    'color_speed': spline(0.5, 0, 1),
    'weight': spline(),
    'opacity': scalespline(max=1),
    'variations': _var_specs,
}

filters = {
    'bilateral': {
        'spatial_std': scalespline(6, d='Spatial filter radius, normalized to 1080p pixels'),
        'color_std': scalespline(0.05, d='Color filter radius, in YUV space'),
        'density_std': scalespline(1.5, d='Density standard deviation'),
        'density_pow': scalespline(0.8, d='Density pre-filter power'),
        'gradient': scalespline(4.0, min=None, d='Intensity of gradient amplification'),
    },
    'colorclip': {
        'gamma': scalespline(4),
        'gamma_threshold': spline(0.01, 0, 1),
        'highlight_power': spline(-1, -1),
        'vibrance': scalespline(),
    },
    'de': {
        'radius': scalespline(11), 'minimum': scalespline(0, max=1), 'curve': scalespline(0.6),
    },
    'haloclip': {},
    'smearclip': {'width': scalespline(0.7, d='Spatial stdev of filter')},
    'plainclip': {'brightness': scalespline(1.0, d='Linear brightness')},
    'logscale': {'brightness': scalespline(4, d='Log-scale brightness')},
    'logencode': {'degamma': scalespline(2.2)},
    'yuv': {},
}

camera = {
    'center': XYPair(spline()),
    'spp': scalespline(d='Samples per pixel multiplier'),
    'dither_width': scalespline(),
    'rotation': spline(period=360),
    'scale': scalespline(),
}

time = {
    'duration': scalar(1),
    'frame_width': scalespline(d='Scale of profile temporal width per frame.'),
}

author = {'name': String(''), 'user': String(''), 'url': String('')}
link = {'src': String(''), 'dst': String('')}
blend = {
    'duration': scalar(2),
    'xform_sort': enum('weightflip weight natural color', 'weightflip'),
    'xform_map': list_(list_(String('xfid'))),
}

base = {
    'name': String('name'), 'base': String('base'),
    'camera': camera, 'filters': filters, 'palette': list_(Palette()),
    'xforms': map_(xform), 'final_xform': xform, 'time': time,
}

node = dict(base, type='node', blend=blend, author=author)
edge = dict(base, type='edge', author=author, blend=blend, link=link,
            xforms=dict(src=map_(xform), dst=map_(xform)))
anim = dict(base, type='animation', authors=list_(author), link=link)

default_filters = ['bilateral', 'logscale', 'smearclip']

# Profile-side filter parameters multiply the genome's splines (cuburn/genome/specs.py:108-112);
# logscale additionally pulls in camera.scale.
prof_filters = dict((fk, dict((k, refscalar(1, '.'.join(['filters', fk, k]))) for k in fv))
                    for fk, fv in filters.items())
prof_filters['logscale']['scale'] = refscalar(1, 'camera.scale')

profile = {
    'duration': RefScalar(30, 'time.duration', 'Base duration in seconds'),
    'fps': Scalar(24, 'Frames per second'),
    'frame_width': refscalar(1, 'time.frame_width'),
    'start': Scalar(None, 'First frame to render (1-indexed, inclusive)'),
    'end': Scalar(None, 'Last frame to render (1-indexed, exclusive)'),
    'skip': Scalar(0, 'Skip this many frames between each rendered frame'),
    'shard': Scalar(0, 'Pack this many frames in each output file'),
    'height': Scalar(720, 'Output height in pixels'),
    'width': Scalar(1280, 'Output width in pixels'),
    'spp': RefScalar(2000, 'camera.spp', 'Base samples per pixel'),
    'filter_order': list_(enum(list(filters.keys())), default_filters),
    'filters': prof_filters,
    'output': {'type': enum('jpeg png tiff x264', 'jpeg')},
}

toplevels = dict(animation=anim, node=node, edge=edge, profile=profile)
