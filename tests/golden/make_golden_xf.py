#!/usr/bin/env python3
"""Golden vectors for the application of an xform — pre affine, every variation, post affine, colour blend — produced by the
REFERENCE's own device-code templates, for tests/test_cpu_golden.py::test_xform_application_matches_reference_templates.

The reference generates the CUDA text of ``apply_xf`` per genome from Tempita templates (cuburn/code/iter.py:81-151,
cuburn/code/variations.py: 95 ``var(name, code, precalc)`` entries) and compiles it with nvcc at run time; neither is possible here.
What IS possible: import those modules (make_golden.prepare_reference: a lib2to3 copy in a temporary directory), render
``iter_xf_body_code`` for a one-xform genome with stand-ins for the packer views of cuburn/code/interp.py (attribute paths become C
identifiers; ``_set`` / ``_code`` collect the precalc blocks that the reference runs in its interpolation kernel), and compile the
rendered text as host C++ with g++ — single-precision libm in place of CUDA's, a three-line restatement of mwc_next / _01 / _11
(cuburn/code/mwc.py:54-76, pinned separately by mwc.json).  Nothing of the rendered text is kept: only inputs (the xform as a genome
dict, points, RNG states) and outputs (points, RNG states, as float32 / uint32 bit patterns) go into xf_apply.json.

    python tests/golden/make_golden_xf.py          (in the build container: needs /root/reference and g++)
"""
import base64
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import make_golden as MG          # noqa: E402

NPTS = 32

PRELUDE = r'''
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <algorithm>
using std::max; using std::min; using std::isfinite; using std::isnan; using std::isinf;
typedef struct { uint32_t mul, state, carry; } mwc_st;
static uint32_t mwc_next(mwc_st &st) { uint64_t t = (uint64_t)st.mul * st.state + st.carry; st.state = (uint32_t)t; st.carry = (uint32_t)(t >> 32); return st.state; }
static float mwc_next_01(mwc_st &st) { return mwc_next(st) * (1.0f / 4294967296.0f); }
static float mwc_next_11(mwc_st &st) { return (float)(int32_t)mwc_next(st) * (1.0f / 2147483648.0f); }
static inline float max(float a, double b) { return fmaxf(a, (float)b); }
static inline float max(double a, float b) { return fmaxf((float)a, b); }
static inline float min(float a, double b) { return fminf(a, (float)b); }
static inline float min(double a, float b) { return fminf((float)a, b); }
#define __device__
'''


class Root(object):
    """What the stand-in views record while the reference's template renders."""
    def __init__(self, xf):
        self.xf, self.used, self.derived, self.precalc = xf, set(), set(), []

    def present(self, path):
        d = self.xf
        for k in path[1:]:
            d = d[k]
        return list(d.keys())


class View(object):
    """Stand-in for the packer views of cuburn/code/interp.py: ``px.pre_affine.angle`` renders as the C identifier
    px_pre_affine_angle; ``_precalc()._set(name)`` names a derived value, ``_code`` takes the block that computes it."""
    def __init__(self, root, path):
        self.__dict__['_r'], self.__dict__['_p'] = root, tuple(path)

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return View(self._r, self._p + (name,))

    def __str__(self):
        self._r.used.add(self._p)
        return '_'.join(self._p)

    def _precalc(self):
        return self

    def _set(self, name):
        self._r.derived.add(self._p + (name,))
        return '_'.join(self._p + (name,))

    def _code(self, code):
        self._r.precalc.append(code)

    def __contains__(self, name):
        return name in self._r.present(self._p)

    def items(self):
        return [(k, View(self._r, self._p + (k,))) for k in self._r.present(self._p)]


def leaf(xf, path):
    d = xf
    for k in path[1:]:
        d = d[k]
    return float(d)


def run_case(ref_iter, stdlib_decls, xf, pts, rng, workdir, tag):
    root = Root(xf)
    body = ref_iter.iter_xf_body(None, '0', View(root, ('px',)))
    inputs = sorted(root.used - root.derived)
    src = [PRELUDE, stdlib_decls]       # (the reference's M_PI ... are single-precision literals, cuburn/code/util.py:141-170)
    for p in inputs:
        src.append('static const float %s = %sf;' % ('_'.join(p), repr(float(np.float32(leaf(xf, p))))))
    for p in sorted(root.derived):
        src.append('static float %s;' % '_'.join(p))
    src.append('static void precalc() {')
    for blk in root.precalc:
        src.append('{' + blk + '}')
    src.append('}')
    src.append(body)
    src.append('static const uint32_t IN[][6] = {')
    for p, r in zip(pts, rng):
        src.append('{%s},' % ', '.join('0x%08xu' % int(v) for v in list(p.view(np.uint32)) + list(r)))
    src.append('};')
    src.append(r'''
int main() {
    precalc();
    for (unsigned i = 0; i < sizeof IN / sizeof *IN; ++i) {
        float x, y, c; mwc_st st = {IN[i][3], IN[i][4], IN[i][5]};
        memcpy(&x, &IN[i][0], 4); memcpy(&y, &IN[i][1], 4); memcpy(&c, &IN[i][2], 4);
        apply_xf_0(x, y, c, st);
        uint32_t o[3]; memcpy(&o[0], &x, 4); memcpy(&o[1], &y, 4); memcpy(&o[2], &c, 4);
        printf("%u %u %u %u %u\n", o[0], o[1], o[2], st.state, st.carry);
    }
    return 0;
}''')
    cpp = os.path.join(workdir, tag + '.cpp')
    exe = os.path.join(workdir, tag)
    open(cpp, 'w').write('\n'.join(src))
    # -O1, no contraction, no fast math: float expressions as written (the CUDA build contracts and uses fast intrinsics:
    # the vectors are the reference's FORMULAS in IEEE single precision, the tolerance of the test covers the rest)
    r = subprocess.run(['g++', '-O1', '-ffp-contract=off', '-fno-fast-math', '-w', '-o', exe, cpp], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('%s: %s' % (tag, r.stderr[:3000]))
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()
    return np.array(out, dtype=np.uint64).astype(np.uint32).reshape(len(pts), 5)


def main():
    tmp, dst = MG.prepare_reference()
    from cuburn.code import iter as ref_iter, util as ref_util
    from cuburn.genome import variations as ref_vars
    from cuburn_amd import configs
    work = tempfile.mkdtemp(prefix='xf_apply_')
    cases = []
    names = [ref_vars.var_names[i] for i in sorted(ref_vars.var_names)]
    rs = np.random.RandomState(606)

    def make_xf(vnames, k0, post):
        vs = {}
        for j, v in enumerate(vnames):
            vd = {'weight': round(0.8 - 0.25 * j, 4)}
            for k, (pn, spec) in enumerate(sorted(ref_vars.var_params[v].items())):
                if pn != 'weight':
                    vd[pn] = round(float(spec.default) + 0.13 * (k + 1) + 0.01 * k0, 4)
            vs[v] = vd
        xf = {'color': round(0.1 + 0.007 * k0, 4), 'color_speed': round(0.2 + 0.005 * k0, 4),
              'pre_affine': configs._affine(20.0 + k0, 0.9, 0.3, 0.2, spread=40.0 + 0.1 * k0, sx=0.85, sy=0.95),
              'variations': vs}
        if post:
            xf['post_affine'] = configs._affine(-10.0, 1.1, -0.1, 0.05, spread=47.0, sx=1.05, sy=1.1)
        return xf

    todo = [([n], i, i % 2 == 0) for i, n in enumerate(names)]
    todo += [(names[i:i + 3], 100 + i, True) for i in (0, 17, 46, 71)]           # sums of several variations in one xform
    for vnames, k0, post in todo:
        xf = make_xf(vnames, k0, post)
        pts = np.zeros((NPTS, 3), np.float32)
        pts[:, 0] = rs.uniform(-1.5, 1.5, NPTS)
        pts[:, 1] = rs.uniform(-1.5, 1.5, NPTS)
        pts[:, 2] = rs.uniform(0, 1, NPTS)
        rng = np.stack([rs.randint(1 << 16, 1 << 32, NPTS, dtype=np.uint64), rs.randint(1, 0x7fffffff, NPTS, dtype=np.uint64),
                        rs.randint(1, 0x7fffffff, NPTS, dtype=np.uint64)], 1).astype(np.uint32)
        out = run_case(ref_iter, ref_util.stdlib.decls, xf, pts, rng, work, 'xf_%03d' % k0)
        b64 = lambda a: base64.b64encode(np.ascontiguousarray(a, '<u4').tobytes()).decode()
        cases.append({'variations': vnames, 'xform': xf,                 # arrays: little-endian uint32, base64, row-major
                      'points_in': b64(pts.view(np.uint32)), 'rng_in': b64(rng),            # [npts][x y colour], [npts][mul state carry]
                      'points_out': b64(out[:, :3]), 'rng_out': b64(out[:, 3:])})           # [npts][x y colour], [npts][state carry]
        print('+'.join(vnames), 'ok')
    with open(os.path.join(HERE, 'xf_apply.json'), 'w') as fp:
        json.dump({'npts': NPTS, 'note': 'float32 / uint32 bit patterns; see make_golden_xf.py', 'cases': cases}, fp, indent=0, sort_keys=True)
    print('wrote xf_apply.json,', len(cases), 'cases')


if __name__ == '__main__':
    main()
