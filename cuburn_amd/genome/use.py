"""
Schema-driven read views of genome / profile documents, and the host spline evaluator.

What the render path needs from cuburn/genome/use.py is small: ``gprof.<attr>`` access that falls
back to the schema's defaults (``gprof.width``, ``gprof.filters.colorclip.gamma(tc)``, ...), profile
scalars that scale a spline of the genome (``gprof.spp(tc)``), and ``SplineEval`` (knot
normalisation for the packer, host evaluation of per-frame scalars).  It is built here as ONE view
class over (document, schema) plus a table of leaf builders, rather than the reference's family of
wrapper subclasses; the behaviour is pinned by tests/golden/{splines,profile,spec_defaults}.json.
"""
import numpy as np

from .spectypes import Enum, Spline, Scalar, RefScalar, Map, List
from .specs import toplevels


class SplineEval(object):
    """Host-side cubic evaluation of one animated parameter (cuburn/genome/use.py:121-185)."""

    def __init__(self, knots, scale, interp='linear'):
        self.knots, self.interp = self.normalize(knots, scale), interp

    @staticmethod
    def normalize(knots, scale):
        """
        JSON spline -> (2, n) array [times; values] (cuburn/genome/use.py:129-158).  A number is a
        constant; ``[a, b]`` runs from a at t=0 to b at t=1 with zero end velocities;
        ``[p0, v0, p1, v1, t2, p2, ...]`` adds end velocities (per unit of genome time, hence
        ``scale``) and interior knots.  Guard knots at t=-2 and t=3 make the Catmull-Rom tangent
        at the first / last real knot equal the requested velocity: each is extrapolated from the
        knot NEXT to the end knot, which is what a centred difference needs.
        """
        if np.isscalar(knots):
            t, v, vel = np.array([0.0, 1.0]), np.array([knots, knots], dtype=np.float64), (0.0, 0.0)
        else:
            flat = np.asarray(knots, dtype=np.float64)
            if flat.size % 2:
                raise ValueError("List with odd number of elements given")
            if flat.size == 2:
                t, v, vel = np.array([0.0, 1.0]), flat.copy(), (0.0, 0.0)
            else:
                t = np.concatenate(([0.0, 1.0], flat[4::2]))
                v = np.concatenate((flat[[0, 2]], flat[5::2]))
                vel = (flat[1], flat[3])
        order = np.lexsort((v, t))                  # by time, ties by value
        t, v = t[order], v[order]
        lead, tail = -2.0, 3.0
        if t[0] >= 0:
            t, v = np.insert(t, 0, lead), np.insert(v, 0, v[1] - (t[1] - lead) * vel[0] * scale)
        if t[-1] <= 1:
            t, v = np.append(t, tail), np.append(v, v[-2] + (tail - t[-2]) * vel[1] * scale)
        return np.stack([t, v])

    def _segment(self, itime):
        """The four knots around ``itime`` with the segment [k1, k2] mapped to [0, 1]:
        (knot times, knot values, local time, 1 / segment length)."""
        kt, kv = self.knots
        first = int(np.clip(np.searchsorted(kt, itime) - 2, 0, kt.size - 4))
        kt, kv = kt[first:first + 4], kv[first:first + 4]
        inv = 1.0 / (kt[2] - kt[1])
        return (kt - kt[1]) * inv, kv, (itime - kt[1]) * inv, inv

    def __call__(self, itime, deriv=0):
        # As in the reference, host evaluation is always linear-domain (use.py:175).
        times, vals, t, scale = self._segment(itime)
        m1 = (vals[2] - vals[0]) / (1.0 - times[0])
        m2 = (vals[3] - vals[1]) / times[3]
        # Hermite basis rows for (m1, p1, m2, p2) in powers t^3, t^2, t, 1
        basis = np.array([[1., -2, 1, 0], [2, -3, 0, 1], [1, -1, 0, 0], [-2, 3, 0, 0]])
        coef = np.array([m1, vals[1], m2, vals[2]]) @ basis
        for _ in range(deriv):
            coef = np.array([0, 3 * coef[0], 2 * coef[1], coef[2]]) * scale
        return float(coef @ np.array([t ** 3, t ** 2, t, 1.0]))

    def scaled(self, factor):
        """The same curve with every knot value multiplied by ``factor`` (a new evaluator)."""
        out = object.__new__(SplineEval)
        out.knots, out.interp = self.knots * np.array([[1.0], [float(factor)]]), self.interp
        return out


def _or_default(value, node):
    return node.default if value is None else value


class View(object):
    """
    Read-only view of ``doc`` under ``schema``.  Attribute (or item) access resolves one child:
    a nested mapping gives another View, a leaf gives its value or the schema's default, and the two
    leaf kinds that need context — animated splines and profile references — are built by
    ``leaves[Spline]`` / ``leaves[RefScalar]`` when given.  Iteration covers the keys that are
    present in the document, sorted as strings (the order the packer lays xforms out in).
    """
    __slots__ = ('_doc', '_schema', '_leaves')

    def __init__(self, doc, schema=None, leaves=None):
        if schema is None:
            kind = doc.get('type')
            if kind not in toplevels:
                raise ValueError('unrecognised document type %r' % (kind,))
            schema = toplevels[kind]
        object.__setattr__(self, '_doc', doc)
        object.__setattr__(self, '_schema', schema)
        object.__setattr__(self, '_leaves', leaves or {})

    def raw(self):
        """The underlying document (only what the file says: no defaults)."""
        return self._doc

    def child_schema(self, key):
        return self._schema.type if isinstance(self._schema, Map) else self._schema[key]

    def _resolve(self, node, value):
        build = self._leaves.get(type(node))
        if build is not None:
            return build(node, value)
        if isinstance(node, (dict, Map)):
            return View(value or {}, node, self._leaves)
        if isinstance(node, List):
            return [self._resolve(node.type, item) for item in _or_default(value, node)]
        if isinstance(node, Enum):
            return value or node.default
        if isinstance(node, (Scalar, RefScalar)):
            return _or_default(value, node)
        return value                                 # splines without a builder, strings, palettes

    def __getattr__(self, key):
        if key.startswith('__'):
            raise AttributeError(key)
        return self._resolve(self.child_schema(key), self._doc.get(key))

    def __getitem__(self, key):
        return getattr(self, str(key))

    def __setattr__(self, key, value):
        raise AttributeError('views are read-only')

    def keys(self):
        return sorted(self._doc)

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._doc)

    def __contains__(self, key):
        self.child_schema(key)                       # unknown names are an error, not "absent"
        return key in self._doc


def genome_view(gnm, scale):
    """Genome whose splines evaluate on the host; ``scale`` = time.duration (velocities are per unit
    of genome time)."""
    return View(gnm, leaves={Spline: lambda node, value: SplineEval(_or_default(value, node), scale, node.interp)})


def profile_view(prof, schema, genome):
    """Profile whose RefScalars are the referenced genome spline times the profile's number
    (cuburn/genome/use.py:100-110): ``gprof.spp(tc)`` = profile spp x genome ``spp`` curve at tc."""
    def ref(node, value):
        target = genome
        for part in node.ref.split('.'):
            target = target[part]
        factor = _or_default(value, node)
        # the referenced leaf is a curve (spp, frame_width, ...) or a plain number (time.duration)
        return target.scaled(factor) if isinstance(target, SplineEval) else target * factor
    return View(prof, schema, leaves={RefScalar: ref})
