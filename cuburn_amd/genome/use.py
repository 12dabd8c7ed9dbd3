"""
Spec-driven views of genome / profile dicts, and the host spline evaluator.

Same surface as cuburn/genome/use.py:6-201: ``Wrapper`` (attribute access with spec
defaults, sorted container protocol), ``RefWrapper`` (profile scalars multiplied into
genome splines), ``SplineWrapper`` and ``SplineEval`` (knot normalisation used by the
packer, host evaluation used for per-frame profile scalars).
"""
import numpy as np
from .spectypes import Enum, Spline, Scalar, RefScalar, Map, List
from .specs import toplevels


class Wrapper(object):
    def __init__(self, val, spec=None, path=(), **params):
        if spec is None:
            assert val.get('type') in toplevels, 'Unrecognized dict type'
            spec = toplevels[val['type']]
        self._val, self.spec, self.path, self._params = val, spec, path, params

    # -- per-type hooks -------------------------------------------------------------
    def wrap(self, name, spec, val):
        path = self.path + (name,)
        if isinstance(spec, Enum):
            return self.wrap_enum(path, spec, val)
        if isinstance(spec, Spline):
            return self.wrap_spline(path, spec, val)
        if isinstance(spec, Scalar):
            return self.wrap_scalar(path, spec, val)
        if isinstance(spec, RefScalar):
            return self.wrap_refscalar(path, spec, val)
        if isinstance(spec, dict):
            return self.wrap_dict(path, spec, val)
        if isinstance(spec, Map):
            return self.wrap_Map(path, spec, val)
        if isinstance(spec, List):
            return self.wrap_List(path, spec, val)
        return val

    def wrap_enum(self, path, spec, val): return val or spec.default
    def wrap_spline(self, path, spec, val): return val
    def wrap_scalar(self, path, spec, val): return val if val is not None else spec.default
    def wrap_refscalar(self, path, spec, val): return val if val is not None else spec.default
    def wrap_dict(self, path, spec, val): return type(self)(val or {}, spec, path, **self._params)
    def wrap_Map(self, path, spec, val): return self.wrap_dict(path, spec, val)

    def wrap_List(self, path, spec, val):
        val = val if val is not None else spec.default
        return [self.wrap(path[-1], spec.type, v) for v in val]

    def get_spec(self, name):
        if isinstance(self.spec, Map):
            return self.spec.type
        return self.spec[name]

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return self.wrap(name, self.get_spec(name), self._val.get(name))

    # -- container protocol: only keys present in the document, sorted ----------------
    def keys(self): return sorted(self._val.keys())
    def items(self): return [(k, self[k]) for k in self.keys()]
    def __iter__(self): return iter(self.keys())
    def __getitem__(self, name): return getattr(self, str(name))

    def __contains__(self, name):
        self.get_spec(name)
        return name in self._val


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

    def find_knots(self, itime):
        kt, kv = self.knots
        idx = int(np.searchsorted(kt, itime)) - 2
        idx = max(0, min(idx, len(kt) - 4))
        times, vals = kt[idx:idx + 4], kv[idx:idx + 4]
        t = itime - times[1]
        times = times - times[1]
        scale = 1 / times[2]
        return times * scale, vals, t * scale, scale

    def __call__(self, itime, deriv=0):
        # As in the reference, host evaluation is always linear-domain (use.py:175).
        times, vals, t, scale = self.find_knots(itime)
        m1 = (vals[2] - vals[0]) / (1.0 - times[0])
        m2 = (vals[3] - vals[1]) / times[3]
        # Hermite basis rows for (m1, p1, m2, p2) in powers t^3, t^2, t, 1
        basis = np.array([[1., -2, 1, 0], [2, -3, 0, 1], [1, -1, 0, 0], [-2, 3, 0, 0]])
        coef = np.array([m1, vals[1], m2, vals[2]]) @ basis
        for _ in range(deriv):
            coef = np.array([0, 3 * coef[0], 2 * coef[1], coef[2]]) * scale
        return float(coef @ np.array([t ** 3, t ** 2, t, 1.0]))

    def __imul__(self, other):
        self.knots[1] *= other
        return self


class SplineWrapper(Wrapper):
    """Genome view whose splines evaluate on the host; needs ``scale`` (= time.duration)."""
    def wrap_spline(self, path, spec, val):
        return SplineEval(val if val is not None else spec.default,
                          self._params['scale'], spec.interp)


class RefWrapper(Wrapper):
    """Profile view: a RefScalar scales the referenced genome spline (use.py:100-110)."""
    def wrap_refscalar(self, path, spec, val):
        spev = self._params['other']
        for part in spec.ref.split('.'):
            spev = spev[part]
        spev *= val if val is not None else spec.default
        return spev
