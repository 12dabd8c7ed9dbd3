"""
flam3 XML flames -> cuburn ``node`` genomes (role of cuburn/genome/convert.py:16-247).

``XMLGenomeParser.parse(text)`` gives one attribute dict per <flame> (with 'xforms',
'palette', optional 'finalxform' / 'symmetry'); ``flam3_to_node`` turns such a dict into a
node document.  Pinned by the reference's own known-answer test
(cuburn/genome/tests/test_convert.py:48-68, reproduced in tests/test_cpu_genome.py).

Conventions that matter (convert.py:113-131): cuburn's IFS space has y pointing the other
way from flam3's, so the affine's cross terms and y offset change sign; an affine is stored
as the bisector ``angle`` of its two axes, their half-opening ``spread``, the axis
``magnitude`` pair and the ``offset``; animated (non-symmetry) xforms rotate one turn per
loop (angle velocity -360).
"""
import binascii
import math
import warnings
import xml.etree.ElementTree as ET

import numpy as np

from . import util
from .variations import var_params
from .blend import node_to_anim, edge_to_anim        # re-exported like the reference (convert.py:13)
from .util import json_encode as to_json             # noqa: F401


class XMLGenomeParser(object):
    """Parse flam3 XML text into a list of flame dicts."""

    def __init__(self):
        self.flames = []

    def feed(self, src):
        root = ET.fromstring(src.strip())
        for el in ([root] if root.tag == 'flame' else root.iter('flame')):
            self.flames.append(self._flame(el))
        return self.flames

    @staticmethod
    def _flame(el):
        flame = dict(el.attrib)
        if el.attrib.get('palette'):
            flame['palette'] = XMLPaletteParser.lookup(int(el.attrib['palette']))
        else:
            flame['palette'] = np.ones((256, 4), dtype=np.float32)
        flame['xforms'] = []
        for child in el:
            attrs = dict(child.attrib)
            if child.tag == 'xform':
                if 'color' in attrs:            # flam3 ignores a second colour coordinate
                    attrs['color'] = attrs['color'].strip().split()[0]
                flame['xforms'].append(attrs)
            elif child.tag == 'finalxform':
                flame['finalxform'] = attrs
            elif child.tag == 'color':
                flame['palette'][int(attrs['index'])][:3] = [float(v) / 255.0 for v in attrs['rgb'].split()]
            elif child.tag == 'symmetry':
                flame['symmetry'] = int(attrs['kind'])
        return flame

    @classmethod
    def parse(cls, src):
        return cls().feed(src)


class XMLPaletteParser(object):
    """flam3-palettes.xml: <palette number= name= data="hex of 256 x 4 bytes">."""
    _names, _numbers = None, None
    _locations = ['/usr/local/share/flam3/flam3-palettes.xml', '/usr/share/flam3/flam3-palettes.xml']

    def __init__(self, src):
        self.names, self.numbers = {}, {}
        for el in ET.fromstring(src.strip()).iter('palette'):
            raw = binascii.a2b_hex(''.join(el.attrib['data'].split()))
            pal = np.frombuffer(raw, np.uint8).reshape((256, 4)) / 255.0
            if 'number' in el.attrib:
                self.numbers[int(el.attrib['number'])] = pal
            if 'name' in el.attrib:
                self.names[el.attrib['name']] = pal

    @classmethod
    def _load(cls):
        for loc in cls._locations:
            try:
                with open(loc) as fp:
                    src = fp.read()
            except IOError:
                continue
            parsed = cls(src)
            cls._names, cls._numbers = parsed.names, parsed.numbers
            return
        raise IOError("Couldn't find a palettes XML file")

    @classmethod
    def lookup(cls, key, isname=False):
        if not cls._names:
            cls._load()
        return np.array((cls._names if isname else cls._numbers)[key])


# ------------------------------------------------------------------ pieces of a node
def convert_affine(aff, animate=False):
    """'xx yx xy yy xo yo' (flam3 coefs order) -> {angle, spread, magnitude, offset}; identity -> None."""
    xx, yx, xy, yy, xo, yo = vals = [float(v) for v in aff.split()]
    if vals == [1, 0, 0, 1, 0, 0]:
        return None
    yx, xy, yo = -yx, -xy, -yo
    x_ang = math.degrees(math.atan2(yx, xx))
    y_ang = math.degrees(math.atan2(yy, xy))
    spread = ((y_ang - x_ang) % 360) / 2
    return dict(spread=spread, angle=(x_ang + spread) % 360,
                magnitude={'x': math.hypot(xx, yx), 'y': math.hypot(xy, yy)},
                offset={'x': xo, 'y': yo})


def convert_vars(xf):
    """Variations present in the xform: weight from attribute <name>, parameters from <name>_<param>."""
    out = {}
    for name, params in var_params.items():
        if name not in xf:
            continue
        var = {'weight': float(xf[name])}
        for p in params:
            if p != 'weight' and name + '_' + p in xf:
                var[p] = float(xf[name + '_' + p])
        out[name] = var
    return out


def convert_xform(xf):
    out = {}
    for dst, attr in (('pre_affine', 'coefs'), ('post_affine', 'post')):
        if attr in xf:
            aff = convert_affine(xf[attr])
            if aff is not None:
                out[dst] = aff
    for key in ('color', 'color_speed', 'opacity', 'weight'):
        if key in xf:
            out[key] = float(xf[key])
    if 'chaos' in xf:
        out['chaos'] = dict(enumerate(float(v) for v in xf['chaos'].split()))
    out['variations'] = convert_vars(xf)
    # the deprecated per-xform 'symmetry' attribute: colour speed, and whether the xform rotates
    symm = float(xf.get('symmetry', 0))
    if 'symmetry' in xf:
        out.setdefault('color_speed', (1 - symm) / 2)
    if xf.get('animate', symm <= 0) and 'pre_affine' in out:
        out['pre_affine']['angle'] = [out['pre_affine']['angle'], -360]
    return out


def make_symm_xforms(kind, offset):
    """The extra xforms of a flame-level <symmetry kind=>: rotations (and a mirror for kind < 0)."""
    assert kind != 0, 'symmetry kind 0 is not a symmetry'
    def plain():
        return dict(color=1, color_speed=0, weight=1, variations={'linear': {'weight': 1}})
    out = []
    if kind < 0:
        out.append(dict(plain(), pre_affine=dict(angle=135, spread=-45)))
        kind = -kind
    for i in range(1, kind):
        xf = plain()
        if kind >= 3:
            xf['color'] = (i - 1) / (kind - 2.0)
        xf['pre_affine'] = dict(angle=(45 + 360 * i / float(kind)) % 360, spread=-45)
        out.append(xf)
    return dict(enumerate(out, offset))


def convert_xforms(flame):
    xfs = dict(enumerate(convert_xform(xf) for xf in flame['xforms']))
    if 'symmetry' in flame:
        xfs.update(make_symm_xforms(flame['symmetry'], len(xfs)))
    return xfs


def flam3_to_node(flame):
    """One parsed flame -> node dict (xform keys are strings '0', '1', ...)."""
    flat = {}

    def put(key, attr, cvt):
        if attr in flame:
            v = cvt(flame[attr])
            if v is not None:
                flat[key] = v

    def pair(v):
        return dict(zip('xy', (float(x) for x in v.split())))

    put('author.name', 'nick', str)
    put('author.url', 'url', lambda s: 'http://' + str(s))
    put('name', 'name', str)
    put('camera.center', 'center', pair)
    put('camera.rotation', 'rotate', float)
    put('camera.dither_width', 'filter', float)
    flat['camera.scale'] = float(flame['scale']) / float(flame['size'].split()[0])
    put('filters.colorclip.gamma', 'gamma', float)
    put('filters.colorclip.gamma_threshold', 'gamma_threshold', float)
    put('filters.colorclip.highlight_power', 'highlight_power', float)
    put('filters.colorclip.vibrance', 'vibrancy', float)
    put('filters.de.curve', 'estimator_curve', float)
    put('filters.de.radius', 'estimator_radius', float)
    if 'estimator_minimum' in flame:
        flat['filters.de.minimum'] = float(flame['estimator_minimum']) / float(flame.get('estimator_radius', 11))
    put('filters.logscale.brightness', 'brightness', float)
    put('palette', 'palette', util.palette_encode)
    flat['xforms'] = convert_xforms(flame)
    put('final_xform', 'finalxform', convert_xform)
    node = util.unflatten(util.flatten(flat))
    node['type'] = 'node'
    return node


def nodes_from_xml_path(path):
    """Every flame of an XML file as a node."""
    with open(path) as fp:
        flames = XMLGenomeParser.parse(fp.read())
    if len(flames) > 10:
        warnings.warn("Lot of flames in this file. Sure it's not a frame-based animation?")
    for flame in flames:
        yield flam3_to_node(flame)
