#!/usr/bin/env python3
"""Golden vectors for the parameter preparation — device-side spline evaluation (linear and magnitude domain), the camera, affine,
density and variation precalc blocks — produced by the REFERENCE's own generated interpolation kernel, for
tests/test_cpu_golden.py::test_parameter_blocks_match_reference_interp_kernel.

``cuburn.code.iter.mkiterlib(genome)`` renders, besides the iterate kernel, the kernel ``interp_iter_params`` that fills one
``iter_params`` struct per temporal sample (cuburn/code/interp.py:234-283 with the precalc hunks the templates registered, and
catmull_rom / catmull_rom_mag of interp.py:284-366 on the knot rows of ``GenomePacker.pack``).  A thread of it depends on nothing but
its index, so the assembled text (taken from the imported reference at generation time, kept nowhere) runs as host C++ behind the
CUDA stand-in of make_golden_filters.py; ``acc_size`` (cuburn/code/iter.py:111-118) gets the reference's calc_dim of the profile.

Kept (interp_params.npz): per genome the struct's field names, the sample times and the float32 blocks of 12 temporal samples.
    python tests/golden/make_golden_interp.py          (in the build container: needs /root/reference and g++)
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import make_golden as MG          # noqa: E402
import make_golden_filters as MF  # noqa: E402

MAIN = r'''
}
int main(int argc, char **argv) {
    FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
    int hdr[8]; float tt[2];
    fread(hdr, 4, 8, in); fread(tt, 4, 2, in);          // rows, row length, samples, width, height, awidth, aheight, astride; tstart, tstep
    const size_t nk = (size_t)hdr[0] * hdr[1];
    std::vector<float> times(nk), knots(nk);
    fread(times.data(), 4, nk, in); fread(knots.data(), 4, nk, in);
    acc_size.width = hdr[3]; acc_size.height = hdr[4]; acc_size.awidth = hdr[5]; acc_size.aheight = hdr[6]; acc_size.astride = hdr[7];
    std::vector<iter_params> P(hdr[2]);
    blockDim = {256, 1, 1}; gridDim = {(unsigned)(hdr[2] + 255) / 256, 1, 1};
    for (int id = 0; id < hdr[2]; ++id) {
        blockIdx = {(unsigned)id / 256, 0, 0}; threadIdx = {(unsigned)id % 256, 0, 0};
        interp_iter_params(P.data(), times.data(), knots.data(), tt[0], tt[1], hdr[2]);
    }
    fwrite(P.data(), sizeof(iter_params), P.size(), out);
    fclose(out);
    return 0;
}
'''
ACC_SIZE = 'typedef struct { uint32_t width, height, awidth, aheight, astride; } acc_size_t;\nstatic acc_size_t acc_size;\n'


def main():
    tmp, dst = MG.prepare_reference()
    from cuburn.code import iter as ref_iter, util
    from cuburn import render
    from cuburn_amd import configs
    work = tempfile.mkdtemp(prefix='interp_ref_')
    out = {}
    NS = 12
    for name in ('cfg3', 'cfg5', 'allvars'):
        gnm, prof = configs.allvars() if name == 'allvars' else configs.CONFIGS[name]()
        packer, lib = ref_iter.mkiterlib(gnm)
        src = util.assemble_code(lib.deps[0]).replace('#include<cuda.h>', '')
        src, n2 = re.subn(r'asm\("cvt\.rni\.s32\.f32\s+%0,\s+%1;" : "=r"\(ret\) : "f"\(f\)\);', 'ret = (uint32_t)(int32_t)rintf(f);', src)
        assert n2 == 1
        times, knots = packer.pack(gnm)
        times, knots = np.ascontiguousarray(times, np.float32), np.ascontiguousarray(knots, np.float32)
        d = render.Framebuffers.calc_dim(prof['width'], prof['height'])
        tstart, tstep = np.float32(0.23), np.float32(0.5 / NS)
        cpp, exe = os.path.join(work, name + '.cpp'), os.path.join(work, name)
        # (max / min of the spline code on int arguments; the struct is declared by the assembled text itself)
        open(cpp, 'w').write(MF.PRELUDE.replace('extern "C" {', 'using std::max; using std::min;\n#include <algorithm>\n' + ACC_SIZE + 'extern "C" {') + src + MAIN)
        r = subprocess.run(['g++', '-O1', '-ffp-contract=off', '-fno-fast-math', '-w', '-o', exe, cpp], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[:4000]
        with open(os.path.join(work, name + '.in'), 'wb') as fp:
            fp.write(np.array([times.shape[0], times.shape[1], NS, d.w, d.h, d.aw, d.ah, d.astride], np.int32).tobytes())
            fp.write(np.array([tstart, tstep], np.float32).tobytes())
            fp.write(times.tobytes()); fp.write(knots.tobytes())
        subprocess.run([exe, os.path.join(work, name + '.in'), os.path.join(work, name + '.out')], check=True)
        blocks = np.fromfile(os.path.join(work, name + '.out'), np.float32).reshape(NS, -1)
        names = ['.'.join(p) for p in packer.packed]
        assert blocks.shape[1] == len(names), (blocks.shape, len(names))
        out[name + '_names'] = np.array(names)
        out[name + '_blocks'] = blocks
        out[name + '_times'] = (tstart + np.arange(NS, dtype=np.float32) * tstep).astype(np.float32)
        out[name + '_dim'] = np.array([d.w, d.h, d.aw, d.ah, d.astride], np.int32)
        print(name, blocks.shape)
    np.savez_compressed(os.path.join(HERE, 'interp_params.npz'), **out)
    print('wrote interp_params.npz')

    # ---- palette interpolation + packing with dither (cuburn/code/interp.py:369-434, host side cuburn/render.py:276-301): the reference's
    # kernel text with its two inline-PTX RNG functions replaced by the three-line restatement pinned by mwc.json, a surface as an array
    from cuburn.code import interp as ref_interp, mwc as ref_mwc
    from cuburn.genome.util import palette_decode
    psrc = util.assemble_code(ref_interp.palintlib).replace('#include<cuda.h>', '')
    psrc, n2 = re.subn(r'asm\("cvt\.rni\.s32\.f32\s+%0,\s+%1;" : "=r"\(ret\) : "f"\(f\)\);', 'ret = (uint32_t)(int32_t)rintf(f);', psrc)
    assert n2 == 1 and ref_mwc.mwclib.defs in psrc
    psrc = psrc.replace(ref_mwc.mwclib.defs, r'''
static uint32_t mwc_next(mwc_st &st) { uint64_t t = (uint64_t)st.mul * st.state + st.carry; st.state = (uint32_t)t; st.carry = (uint32_t)(t >> 32); return st.state; }
static float mwc_next_01(mwc_st &st) { return mwc_next(st) * (1.0f / 4294967296.0f); }
static float mwc_next_11(mwc_st &st) { return (float)(int32_t)mwc_next(st) * (1.0f / 2147483648.0f); }
''')
    pal_prelude = MF.PRELUDE.replace('extern "C" {', r'''
#include <algorithm>
using std::max; using std::min;
struct uint2 { uint32_t x, y; };
enum { cudaSurfaceType2D = 2 };
template <typename T, int D> struct surface { uint2 *data; int w; };
static inline void surf2Dwrite(uint2 v, surface<void, 2> &s, int xbytes, int y) { s.data[(size_t)y * s.w + xbytes / 8] = v; }
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
static inline uint32_t min(int a, uint32_t b) { return (uint32_t)a < b ? (uint32_t)a : b; }
extern "C" {''')
    pal_main = r'''
}
int main(int argc, char **argv) {
    FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
    int hdr[3]; float tt[2];
    fread(hdr, 4, 3, in); fread(tt, 4, 2, in);          // palettes, rows, rng states; tstart, tstep
    std::vector<float> times(32); std::vector<float4> src((size_t)hdr[0] * 256); std::vector<mwc_st> rng(hdr[2]);
    fread(times.data(), 4, 32, in); fread(src.data(), 16, src.size(), in); fread(rng.data(), 12, rng.size(), in);
    std::vector<uint2> pal((size_t)hdr[1] * 256);
    flatpal.data = pal.data(); flatpal.w = 256;
    ringbuf rb = {0, 0};
    blockDim = {256, 1, 1}; gridDim = {(unsigned)hdr[1], 1, 1};
    for (int b = 0; b < hdr[1]; ++b) for (int t = 0; t < 256; ++t) {          // (thread 0 of a block first: it draws the block's ring-buffer slot)
        blockIdx = {(unsigned)b, 0, 0}; threadIdx = {(unsigned)t, 0, 0};
        interp_palette_flat(&rb, rng.data(), times.data(), src.data(), tt[0], tt[1]);
    }
    fwrite(pal.data(), 8, pal.size(), out); fwrite(rng.data(), 12, rng.size(), out);
    fclose(out);
    return 0;
}
'''
    cpp, exe = os.path.join(work, 'pal.cpp'), os.path.join(work, 'pal')
    open(cpp, 'w').write(pal_prelude + psrc + pal_main)
    r = subprocess.run(['g++', '-O1', '-ffp-contract=off', '-fno-fast-math', '-w', '-o', exe, cpp], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:4000]
    pout = {}
    for name, ts, td in (('cfg3', 0.21, 0.04), ('cfg2', 0.5, 0.0), ('allvars', 0.9, 0.2)):
        gnm, prof = configs.allvars() if name == 'allvars' else configs.CONFIGS[name]()
        palsrc = dict([(v[0], palette_decode(v[1:])) for v in gnm['palette']])
        ptimes, pvals = zip(*sorted(palsrc.items()))
        palettes = np.array(pvals, np.float32)
        ptime = np.full(32, 1e9, np.float32); ptime[:len(ptimes)] = ptimes
        rows = 64
        seeds = ref_mwc.make_seeds(rows * 256, host_seed=1234)
        with open(os.path.join(work, 'pal.in'), 'wb') as fp:
            fp.write(np.array([len(ptimes), rows, len(seeds)], np.int32).tobytes())
            fp.write(np.array([ts, td / rows], np.float32).tobytes())
            fp.write(ptime.tobytes()); fp.write(palettes.tobytes()); fp.write(np.ascontiguousarray(seeds, np.uint32).tobytes())
        subprocess.run([exe, os.path.join(work, 'pal.in'), os.path.join(work, 'pal.out')], check=True)
        raw = np.fromfile(os.path.join(work, 'pal.out'), np.uint32)
        pal = raw[:rows * 256 * 2].reshape(rows, 256, 2)
        packed = np.ascontiguousarray(pal[..., 0].astype(np.uint64) | (pal[..., 1].astype(np.uint64) << np.uint64(32)))
        rng_after = np.ascontiguousarray(raw[rows * 256 * 2:].reshape(-1, 3))
        # bit-exact quantities: digests of the whole arrays, four rows in full for a failing test to look at
        pout[name + '_packed_sha256'] = np.array(hashlib.sha256(packed.tobytes()).hexdigest())
        pout[name + '_rng_after_sha256'] = np.array(hashlib.sha256(rng_after.tobytes()).hexdigest())
        pout[name + '_packed_rows'] = packed[[0, 1, 31, 63]]
        pout[name + '_ts_td'] = np.array([ts, td], np.float32)
    pout['host_seed'] = np.int32(1234)
    np.savez_compressed(os.path.join(HERE, 'interp_palette.npz'), **pout)
    print('wrote interp_palette.npz')


if __name__ == '__main__':
    main()
