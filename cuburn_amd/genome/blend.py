"""
Node / edge genomes -> animation genomes.

The renderer only consumes ``animation`` documents (splines over t in [0, 1]).  A still
``node`` (every value a position or a [position, velocity] pair) becomes a looping
animation by blending the node with a time-shifted copy of itself; an ``edge`` blends two
different nodes.  Behaviour follows cuburn/genome/blend.py:16-311 (node_to_anim :16-25,
edge_to_anim :27-33, resolve :35-59, temporal offset :67-79, blend :81-136, spline
construction :160-203, xform padding :239-272, xform pairing :280-331); the reference's
palette-flip helpers (:333-358) reference undefined names there and are not part of any
call path, so they have no counterpart here.

The reference is Python 2: where its behaviour rests on py2 semantics (round() half away
from zero, ordering of mixed int/list/str values) the helpers below spell that out.
"""
import math
from itertools import zip_longest

from . import specs, spectypes, variations
from .util import flatten, get, resolve_spec, unflatten


# ------------------------------------------------------------------ py2 semantics
def _round(x):
    """Python 2 round(): halves away from zero (Python 3 rounds halves to even)."""
    return math.copysign(math.floor(abs(x) + 0.5), x)


def _mixed_key(v):
    """Python 2 ordered numbers before lists before strings."""
    if isinstance(v, (int, float)):
        return (0, v, ())
    if isinstance(v, (list, tuple)):
        return (1, 0, tuple(v))
    return (2, 0, (str(v),))


def _wide(xf, which):
    """spread > 90 test of blend.py:245-248,299-300; a [pos, vel] list compares greater (py2)."""
    v = get(xf, 45, which, 'spread')
    return True if isinstance(v, (list, tuple)) else v > 90


# ------------------------------------------------------------------ entry points
def node_to_anim(gdb, node, half):
    """A loop of one node: t in [0,1] covers one period (or the middle half period if ``half``)."""
    node = resolve(gdb, node)
    t0, t1 = (-0.25, 0.25) if half else (0, 1)
    edge = {'blend': {'duration': t1 - t0, 'xform_sort': 'natural'}}
    return blend(apply_temporal_offset(node, t0), apply_temporal_offset(node, t1), edge)


def edge_to_anim(gdb, edge):
    edge = resolve(gdb, edge)
    ends = []
    for side in ('src', 'dst'):
        ident, offset = _split_ref_id(edge['link'][side])
        ends.append(apply_temporal_offset(resolve(gdb, gdb.get(ident)), offset))
    return blend(ends[0], ends[1], edge)


def _split_ref_id(ref):
    """'node_id@0.5' -> ('node_id', 0.5); no offset -> 0."""
    ident, _, offset = ref.partition('@')
    return ident, (float(offset) if offset else 0)


def resolve(gdb, item):
    """
    Merge an item with its chain of ``base`` items (oldest first, later wins).  In edges,
    spline and list values of the chain are concatenated instead (they are edit lists).
    """
    top = specs.toplevels[item['type']]
    chain = [item]
    while chain[0].get('base') is not None:
        chain.insert(0, gdb.get(chain[0]['base']))
    flat = [flatten(i) for i in chain]
    concat = item['type'] == 'edge'
    out = {}
    for key in set(k for f in flat for k in f):
        vals = [f[key] for f in flat if key in f]
        sp = resolve_spec(top, key.split('.'))
        if concat and isinstance(sp, (spectypes.Spline, spectypes.List)):
            out[key] = [x for v in vals for x in v]
        else:
            out[key] = vals[-1]
    return unflatten(out)


def apply_temporal_offset(node, offset=0):
    """Advance every periodic [position, velocity] spline of a node by ``offset`` time units."""
    def walk(spec, val):
        if isinstance(spec, spectypes.Spline):
            if spec.period is not None and isinstance(val, list) and val[1]:
                return [val[0] + offset * val[1], val[1]]
            return val
        if isinstance(spec, spectypes.Map) and isinstance(val, dict):
            return dict((k, walk(spec.type, v)) for k, v in val.items())
        if isinstance(spec, dict) and isinstance(val, dict):
            return dict((k, walk(spec[k], v)) for k, v in val.items())      # KeyError: not in schema
        if isinstance(spec, spectypes.List) and isinstance(val, list):
            return [walk(spec.type, v) for v in val]
        return val
    return walk(specs.toplevels[node['type']], node)


# ------------------------------------------------------------------ blending
def blend(src, dst, edit={}):
    """
    Blend two resolved, offset-adjusted nodes into an animation dict.  ``edit`` is the
    resolved edge (may be empty).
    """
    raw = {}
    for d in (src, dst, edit):
        raw.update(d.get('blend', {}))
    duration = raw.get('duration', specs.blend['duration'].default)
    sort = raw.get('xform_sort') or specs.blend['xform_sort'].default
    explicit = raw.get('xform_map') or []

    out = merge_nodes(specs.node, src, dst, edit, duration)
    out['xforms'] = {}
    for skey, dkey in sort_xforms(src['xforms'], dst['xforms'], sort, explicit):
        edits = merge_edits(specs.xform, get(edit, {}, 'xforms', 'src', skey),
                            get(edit, {}, 'xforms', 'dst', dkey))
        # a 'dup' partner fades the (padded) copy in or out through its weight
        if skey == 'dup':
            edits.setdefault('weight', []).extend([0, 0])
        if dkey == 'dup':
            edits.setdefault('weight', []).extend([1, 0])
        out['xforms'][(skey or 'pad') + '_' + (dkey or 'pad')] = blend_xform(
            src['xforms'].get(skey), dst['xforms'].get(dkey), edits, duration)

    if 'final_xform' in src or 'final_xform' in dst:
        out['final_xform'] = blend_xform(src.get('final_xform'), dst.get('final_xform'),
                                         edit.get('final_xform'), duration, True)
    out['type'] = 'animation'
    out.setdefault('time', {})['duration'] = duration
    return out


def merge_edits(sv, av, bv):
    """Combine the src-side and dst-side edit trees of one xform: lists append, scalars: b wins."""
    if isinstance(sv, (dict, spectypes.Map)):
        av, bv = av or {}, bv or {}
        sub = (lambda k: sv.type) if isinstance(sv, spectypes.Map) else (lambda k: sv[k])
        return dict((k, merge_edits(sub(k), av.get(k), bv.get(k))) for k in set(av) | set(bv))
    if isinstance(sv, (spectypes.List, spectypes.Spline)):
        return (av or []) + (bv or [])
    return bv if bv is not None else av


def _pos_vel(spl, val):
    if val is None:
        return spl.default, 0
    if isinstance(val, (int, float)):
        return val, 0
    return val[0], val[1]


def tospline(spl, src, dst, edit, duration):
    """
    One animated value from the two node values (number or [pos, vel]) and the edge's knot
    edits [t, v, t, v, ...]: a constant, a [v0, v1] ramp, or [p0, v0, p1, v1, extra knots...].
    """
    sp, sv = _pos_vel(spl, src)
    dp, dv = _pos_vel(spl, dst)
    if spl.var:                 # variation parameters hold their value through a missing side
        if src is None:
            sp = dp
        if dst is None:
            dp = sp

    knots = dict(zip(edit[::2], edit[1::2])) if edit else {}
    e0, e1 = knots.pop(0, None), knots.pop(1, None)
    extra = [x for k, v in knots.items() if v is not None for x in (k, v)]

    if spl.period:
        # choose the number of whole turns from the mean end velocity, keep dp congruent
        period = spl.period
        turns = duration * (sv + dv) / (2.0 * period)
        frac = (float(dp - sp) / period) % (1.0 if turns >= 0 else -1.0)
        dp = sp + (_round(turns - frac) + frac) * period
        # explicit end knots pick the nearest congruent value
        if e0 is not None:
            sp += _round(float(e0 - sp) / period) * period
        if e1 is not None:
            dp += _round(float(e1 - dp) / period) * period
    if extra or sv or dv or e0 or e1:
        return [sp, sv, dp, dv] + extra
    if sp != dp:
        return [sp, dp]
    return sp


def merge_nodes(sp, src, dst, edit, duration):
    """Walk the schema; splines become animated values, lists append, anything else: last wins."""
    if isinstance(sp, dict):
        src, dst, edit = src or {}, dst or {}, edit or {}
        return dict((k, merge_nodes(sp[k], src.get(k), dst.get(k), edit.get(k), duration))
                    for k in set(src) | set(dst) | set(edit) if k in sp)
    if isinstance(sp, spectypes.Spline):
        return tospline(sp, src, dst, edit, duration)
    if isinstance(sp, spectypes.List):
        if isinstance(sp.type, spectypes.Palette):      # palettes get their time stamp
            src = [[0] + src] if src is not None else None
            dst = [[1] + dst] if dst is not None else None
        return (src or []) + (dst or []) + (edit or [])
    return edit if edit is not None else dst if dst is not None else src


def blend_xform(sxf, dxf, edits, duration, isfinal=False):
    if sxf is None:
        sxf = padding_xform(dxf, isfinal)
    if dxf is None:
        dxf = padding_xform(sxf, isfinal)
    return merge_nodes(specs.xform, sxf, dxf, edits, duration)


# a partner for an unpaired xform: close to an identity the xform can morph into
hole_variations = 'spherical ngon julian juliascope polar wedge_sph wedge_julia bipolar'.split()
ident_variations = 'rectangles fan2 blob perspective super_shape'.split()


def padding_xform(xf, isfinal):
    out = {'variations': {}, 'pre_affine': {'angle': 45}}
    if isfinal:
        out.update(weight=0, color_speed=0)
    if _wide(xf, 'pre_affine'):
        out['pre_affine'] = {'angle': 135, 'spread': 135}
    if _wide(xf, 'post_affine'):
        out['post_affine'] = {'angle': 135, 'spread': 135}
    for name in xf.get('variations', {}):
        if name in hole_variations:
            # these blow up around the origin: use the inverted identity instead
            out['pre_affine']['angle'] += 180
            out['variations'] = {'linear': {'weight': -1}}
            return out
        if name in ident_variations:
            out['variations'][name] = dict((p, d) for p, (d, _) in variations.var_params[name].items())
    if out['variations']:
        for v in out['variations'].values():
            v['weight'] = 1.0 / len(out['variations'])
    else:
        out['variations']['linear'] = {'weight': 1}
    return out


def sort_xforms(sxfs, dxfs, sortmethod, explicit=[]):
    """
    Pair up the xforms of the two nodes: explicit pairs first (later entries displace
    earlier ones that name the same xform), then by class (flipped pre / post affine) and
    rank within the class.  Yields (src key or None, dst key or None).
    """
    fwd, rev = {}, {}
    for s, d in explicit:
        if s not in ('pad', 'dup') and s in fwd:
            rev.pop(fwd.pop(s, None), None)
        if d not in ('pad', 'dup') and d in rev:
            fwd.pop(rev.pop(d, None), None)
        fwd[s] = d
        rev[d] = s
    for pair in sorted(fwd.items()):
        yield pair

    def classes(xfs, taken):
        cl = {}
        for k, v in xfs.items():
            if k not in taken:
                cl.setdefault((_wide(v, 'pre_affine'), _wide(v, 'post_affine')), []).append(k)
        return cl
    scl, dcl = classes(sxfs, fwd), classes(dxfs, rev)

    def rank(keys, xfs):
        if sortmethod in ('weight', 'weightflip'):
            return sorted(keys, key=lambda k: _mixed_key(xfs[k].get('weight', 0)))
        if sortmethod == 'color':
            return sorted(keys, key=lambda k: _mixed_key(xfs[k].get('color', 0)))
        return sorted(keys, key=lambda k: (0, int(k), '') if _is_int(k) else (1, 0, k))

    for cl in sorted(set(scl) | set(dcl)):
        ss, ds = rank(scl.get(cl, []), sxfs), rank(dcl.get(cl, []), dxfs)
        if sortmethod == 'weightflip':
            ds = ds[::-1]
        for pair in zip_longest(ss, ds):
            yield pair


def _is_int(k):
    try:
        int(k)
        return True
    except ValueError:
        return False
