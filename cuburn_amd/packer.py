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

``packed`` lists the path of every word of the parameter block, in block order, like the
reference's ``packer.packed`` (plus ``pad`` entries for alignment and ``#id`` / ``#nvar``
entries for the integer structure words the fixed-stride layout carries).
"""
import numpy as np

from .genome import specs
from .genome.use import SplineEval
from .genome.util import resolve_spec
from .genome import variations as V

KNOTS = 32          # 1 << DEFAULT_SEARCH_ROUNDS, cuburn/code/util.py:235
PROG_MAGIC = 0x464c5032
OP_SPLINE, OP_SPLINE_MAG, OP_CAMERA, OP_AFFINE, OP_CDF, OP_RATIO2, OP_INVSQ, OP_PERSP, OP_INVSQ_MAX, OP_CONST = range(10)
MAX_PSTRIDE = 4096
XF_HDR = 16

_AFFINE_ROWS = (('angle',), ('spread',), ('magnitude', 'x'), ('magnitude', 'y'), ('offset', 'x'), ('offset', 'y'))
_AFFINE_OUT = ('xx', 'xy', 'xo', 'yx', 'yy', 'yo')
_CAMERA_ROWS = (('rotation',), ('center', 'x'), ('center', 'y'), ('scale',))


class GenomePacker(object):
    def __init__(self, gnm):
        self.rows = []      # [(path, is_mag)]
        self.ops = []       # [(kind, dst, a, b)]
        self.packed = []    # [path] per block word ('pad' = unused, '#...' = integer structure word)
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

    def _spline_op(self, dst, path):
        r = self._row(path)
        self.ops.append((OP_SPLINE_MAG if self.rows[r][1] else OP_SPLINE, dst, r, 0))

    def _affine(self, base, dst):
        for i, o in enumerate(_AFFINE_OUT):
            self.packed[dst + i] = tuple(base) + (o,)
        first = self._new_rows(base, _AFFINE_ROWS)
        self.ops.append((OP_AFFINE, dst, first, 0))
        return first

    def _xform(self, base, xf, rec):
        """Fill the fixed-stride record starting at block offset ``rec`` (include/flame_hip.h (5))."""
        pre_rows = self._affine(base + ('pre_affine',), rec)
        has_post = 1 if 'post_affine' in xf else 0
        if has_post:
            self._affine(base + ('post_affine',), rec + 6)
        self.packed[rec + 12] = base + ('color',)
        self.packed[rec + 13] = base + ('color_speed',)
        self._spline_op(rec + 12, base + ('color',))
        self._spline_op(rec + 13, base + ('color_speed',))
        names = sorted(xf.get('variations', {}))
        self.packed[rec + 14] = base + ('#nvar',)
        self.ops.append((OP_CONST, rec + 14, len(names) | (has_post << 8), 0))
        for j, vname in enumerate(names):
            vbase = base + ('variations', vname)
            voff = rec + XF_HDR + j * self.var_stride
            layout = V.record_layout(vname)
            self.packed[voff] = vbase + ('#id',)
            self.ops.append((OP_CONST, voff, V.var_ids[vname], 0))
            self.packed[voff + 1] = vbase + ('weight',)
            self._spline_op(voff + 1, vbase + ('weight',))
            for i, n in enumerate(layout):
                self.packed[voff + 2 + i] = vbase + (n,)
            direct = sorted(k for k in V.var_params[vname] if k != 'weight')
            for i, pname in enumerate(direct):
                self._spline_op(voff + 2 + i, vbase + (pname,))
            dst = voff + 2 + len(direct)
            for pname, kind, src in V.var_precalc.get(vname, ()):
                if kind == 'invsq':      # source lives on the xform's pre affine
                    r = pre_rows + [tuple(x) for x in _AFFINE_ROWS].index(tuple(src.split('.')[1:]))
                    self.ops.append((OP_INVSQ, dst, r, 0)); dst += 1
                elif kind == 'invsq_max':
                    self.ops.append((OP_INVSQ_MAX, dst, self._row(vbase + (src,)), 0)); dst += 1
                elif kind == 'ratio2':
                    self.ops.append((OP_RATIO2, dst, self._row(vbase + (src[0],)), self._row(vbase + (src[1],)))); dst += 1
                elif kind == 'persp':
                    self.ops.append((OP_PERSP, dst, self._row(vbase + (src[0],)), self._row(vbase + (src[1],)))); dst += 3

    def _build(self, gnm):
        xforms = gnm.get('xforms', {})
        keys = sorted(xforms.keys())          # string sort, cuburn/genome/use.py:88-91
        if not keys:
            raise ValueError('genome has no xforms')
        has_final = 1 if 'final_xform' in gnm else 0
        self.xform_keys = keys
        allxf = [(('xforms', k), xforms[k]) for k in keys]
        if has_final:
            allxf.append((('final_xform',), gnm['final_xform']))
        for _, xf in allxf:
            for vname in xf.get('variations', {}):
                if vname not in V.var_ids:
                    raise ValueError('unknown variation %r' % vname)
        # strides: every xform record and every variation record has the same length
        maxv = max([len(xf.get('variations', {})) for _, xf in allxf] + [1])
        maxp = max([len(V.record_layout(v)) for _, xf in allxf for v in xf.get('variations', {})] + [0])
        self.var_stride = 2 + maxp
        self.xf_stride = (XF_HDR + maxv * self.var_stride + 3) // 4 * 4
        cdf = 6
        self.xf_off = (cdf + len(keys) + 3) // 4 * 4
        total = self.xf_off + len(allxf) * self.xf_stride
        self.packed = [('pad', str(i)) for i in range(total)]
        # camera
        for i, o in enumerate(_AFFINE_OUT):
            self.packed[i] = ('camera', o)
        first = self._new_rows(('camera',), _CAMERA_ROWS)
        self.ops.append((OP_CAMERA, 0, first, 0))
        # cumulative xform densities
        for i, k in enumerate(keys):
            self.packed[cdf + i] = ('den', k)
        first = self._new_rows((), [('xforms', k, 'weight') for k in keys])
        self.ops.append((OP_CDF, cdf, first, len(keys)))
        for i, (base, xf) in enumerate(allxf):
            self._xform(base, xf, self.xf_off + i * self.xf_stride)
        self._prog = [PROG_MAGIC, len(keys), has_final, 0, cdf, self.xf_off, self.xf_stride, self.var_stride]

    # ------------------------------------------------------------------ per-frame data
    def signature(self, gnm):
        """
        Hashable snapshot of every spline this genome structure reads from ``gnm`` (by value, so
        in-place edits of the document are seen), plus the time scale.  Frames of an animation
        share it: the device evaluates the splines at each frame's times, the knots do not change.
        """
        sig = [gnm.get('time', {}).get('duration', 1)]
        for path, _ in self.rows:
            attr = gnm
            for name in path:
                if not isinstance(attr, dict) or name not in attr:
                    attr = None
                    break
                attr = attr[name]
            sig.append(tuple(attr) if isinstance(attr, (list, tuple)) else attr)
        return tuple(sig)

    def pack(self, gnm, pool=None, sig=None):
        """
        Knot times and values for every row, as two float32 arrays of shape (nrows, 32);
        times padded with 1e9 (cuburn/code/interp.py:207-232).  Padding of ``knots`` is 0
        (the reference leaves it uninitialised).  The result for the most recent signature is
        kept: packing 100 splines costs milliseconds of numpy on the host, a frame of the hot
        path takes about as long on the device.
        """
        sig = self.signature(gnm) if sig is None else sig
        if getattr(self, '_packed_sig', None) == sig:
            return self._packed
        times = np.full((self.nrows, KNOTS), 1e9, dtype=np.float32)
        knots = np.zeros((self.nrows, KNOTS), dtype=np.float32)
        scale = sig[0]
        for idx, (path, _) in enumerate(self.rows):
            attr = sig[idx + 1]
            if attr is None:
                attr = resolve_spec(specs.anim, path).default
            kt = SplineEval.normalize(list(attr) if isinstance(attr, tuple) else attr, scale)
            n = kt.shape[1]
            if n > KNOTS:
                raise ValueError('spline %s has %d knots (max %d)' % ('.'.join(path), n, KNOTS))
            times[idx, :n] = kt[0]
            knots[idx, :n] = kt[1]
        self._packed_sig, self._packed = sig, (times, knots)
        return times, knots
