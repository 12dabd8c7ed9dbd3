"""
GenomePacker: decides the parameter-block layout for a genome and packs its splines.

Role of cuburn/code/interp.py:125-282.  The reference discovers the layout as a side
effect of rendering CUDA templates; here the genome's *structure* (which xforms, which
variations, post affines, final xform) is turned into three plain arrays handed to
libflame_hip (formats: include/flame_hip.h (4)-(6)):

  * ``prog``  int32 xform program interpreted by the iterate kernel,
  * ``ops``   int32 x4 interpolation ops evaluated per temporal sample on device,
  * rows      one spline (32 knot times + 32 knot values) per genome parameter,
              filled per frame by ``pack`` exactly as GenomePacker.pack (interp.py:207-232).

``packed`` lists the path of every float of the parameter block, in block order, like the
reference's ``packer.packed``.
"""
import numpy as np

from .genome import specs
from .genome.use import SplineEval
from .genome.util import resolve_spec
from .genome import variations as V

KNOTS = 32          # 1 << DEFAULT_SEARCH_ROUNDS, cuburn/code/util.py:235
PROG_MAGIC = 0x464c5031
OP_SPLINE, OP_SPLINE_MAG, OP_CAMERA, OP_AFFINE, OP_CDF, OP_RATIO2, OP_INVSQ, OP_PERSP, OP_INVSQ_MAX = range(9)
MAX_PSTRIDE = 1024

_AFFINE_ROWS = (('angle',), ('spread',), ('magnitude', 'x'), ('magnitude', 'y'), ('offset', 'x'), ('offset', 'y'))
_AFFINE_OUT = ('xx', 'xy', 'xo', 'yx', 'yy', 'yo')
_CAMERA_ROWS = (('rotation',), ('center', 'x'), ('center', 'y'), ('scale',))


class GenomePacker(object):
    def __init__(self, gnm):
        self.rows = []      # [(path, is_mag)]
        self.ops = []       # [(kind, dst, a, b)]
        self.packed = []    # [path] per block float
        self._row_index = {}
        self._build(gnm)
        self.genome = [p for p, _ in self.rows]
        self.nrows = len(self.rows)
        self.pstride = len(self.packed)
        if self.pstride > MAX_PSTRIDE:
            raise ValueError('genome needs %d parameter floats (max %d)' % (self.pstride, MAX_PSTRIDE))
        self.prog = np.array(self._prog, dtype=np.int32)
        self.prog[3] = self.pstride
        self.ops_array = np.array(self.ops, dtype=np.int32).reshape(-1, 4)

    def __len__(self):
        """Length of the parameter block in floats (reference: len(packer))."""
        return self.pstride

    # ------------------------------------------------------------------ layout
    def _row(self, path):
        path = tuple(path)
        if path not in self._row_index:
            spec = resolve_spec(specs.anim, path)
            self._row_index[path] = len(self.rows)
            self.rows.append((path, spec.interp == 'mag'))
        return self._row_index[path]

    def _new_rows(self, base, subpaths):
        """Allocate consecutive fresh rows (ops that take a row range need contiguity)."""
        first = len(self.rows)
        for sp in subpaths:
            path = tuple(base) + tuple(sp)
            spec = resolve_spec(specs.anim, path)
            self.rows.append((path, spec.interp == 'mag'))
            self._row_index.setdefault(path, len(self.rows) - 1)
        return first

    def _alloc(self, names):
        off = len(self.packed)
        self.packed.extend(tuple(n) for n in names)
        return off

    def _spline_op(self, dst, path):
        r = self._row(path)
        self.ops.append((OP_SPLINE_MAG if self.rows[r][1] else OP_SPLINE, dst, r, 0))

    def _affine(self, base):
        dst = self._alloc(base + (o,) for o in _AFFINE_OUT)
        first = self._new_rows(base, _AFFINE_ROWS)
        self.ops.append((OP_AFFINE, dst, first, 0))
        return dst, first

    def _xform(self, base, xf):
        poff, pre_rows = self._affine(base + ('pre_affine',))
        flags = 0
        if 'post_affine' in xf:
            flags |= 1
            self._affine(base + ('post_affine',))
        c = self._alloc([base + ('color',), base + ('color_speed',)])
        self._spline_op(c, base + ('color',))
        self._spline_op(c + 1, base + ('color_speed',))
        desc = [poff, flags, 0]
        for vname in sorted(xf.get('variations', {})):
            if vname not in V.var_ids:
                raise ValueError('unknown variation %r' % vname)
            vbase = base + ('variations', vname)
            layout = V.record_layout(vname)
            voff = self._alloc([vbase + ('weight',)] + [vbase + (n,) for n in layout])
            self._spline_op(voff, vbase + ('weight',))
            direct = sorted(k for k in V.var_params[vname] if k != 'weight')
            for i, pname in enumerate(direct):
                self._spline_op(voff + 1 + i, vbase + (pname,))
            dst = voff + 1 + len(direct)
            for pname, kind, src in V.var_precalc.get(vname, ()):
                if kind == 'invsq':      # source lives on the xform's pre affine
                    r = pre_rows + [tuple(s) for s in _AFFINE_ROWS].index(tuple(src.split('.')[1:]))
                    self.ops.append((OP_INVSQ, dst, r, 0)); dst += 1
                elif kind == 'invsq_max':
                    self.ops.append((OP_INVSQ_MAX, dst, self._row(vbase + (src,)), 0)); dst += 1
                elif kind == 'ratio2':
                    self.ops.append((OP_RATIO2, dst, self._row(vbase + (src[0],)), self._row(vbase + (src[1],)))); dst += 1
                elif kind == 'persp':
                    self.ops.append((OP_PERSP, dst, self._row(vbase + (src[0],)), self._row(vbase + (src[1],)))); dst += 3
            desc[2] += 1
            desc.extend([V.var_ids[vname], voff])
        return desc

    def _build(self, gnm):
        xforms = gnm.get('xforms', {})
        keys = sorted(xforms.keys())          # string sort, cuburn/genome/use.py:88-91
        if not keys:
            raise ValueError('genome has no xforms')
        has_final = 1 if 'final_xform' in gnm else 0
        self.xform_keys = keys
        # camera
        cam = self._alloc(('camera', o) for o in _AFFINE_OUT)
        first = self._new_rows(('camera',), _CAMERA_ROWS)
        self.ops.append((OP_CAMERA, cam, first, 0))
        # cumulative xform densities
        cdf = self._alloc(('den', k) for k in keys)
        first = self._new_rows((), [('xforms', k, 'weight') for k in keys])
        self.ops.append((OP_CDF, cdf, first, len(keys)))
        descs = [self._xform(('xforms', k), xforms[k]) for k in keys]
        if has_final:
            descs.append(self._xform(('final_xform',), gnm['final_xform']))
        prog = [PROG_MAGIC, len(keys), has_final, 0, cdf, 0, 0, 0]
        off = len(prog) + len(descs)
        table = []
        for d in descs:
            table.append(off)
            off += len(d)
        prog.extend(table)
        for d in descs:
            prog.extend(d)
        self._prog = prog

    # ------------------------------------------------------------------ per-frame data
    def pack(self, gnm, pool=None):
        """
        Knot times and values for every row, as two float32 arrays of shape (nrows, 32);
        times padded with 1e9 (cuburn/code/interp.py:207-232).  Padding of ``knots`` is 0
        (the reference leaves it uninitialised).
        """
        times = np.full((self.nrows, KNOTS), 1e9, dtype=np.float32)
        knots = np.zeros((self.nrows, KNOTS), dtype=np.float32)
        scale = gnm.get('time', {}).get('duration', 1)
        for idx, (path, _) in enumerate(self.rows):
            attr = gnm
            for name in path:
                if not isinstance(attr, dict) or name not in attr:
                    attr = resolve_spec(specs.anim, path).default
                    break
                attr = attr[name]
            kt = SplineEval.normalize(attr, scale)
            n = kt.shape[1]
            if n > KNOTS:
                raise ValueError('spline %s has %d knots (max %d)' % ('.'.join(path), n, KNOTS))
            times[idx, :n] = kt[0]
            knots[idx, :n] = kt[1]
        return times, knots
