#!/usr/bin/env python3
"""Golden vectors for the output conversion — rgba8 / rgba16 and the planar YUV formats of the video outputs, with dither — produced by
the REFERENCE's own CUDA text (cuburn/code/output.py ``pixfmtlib``) compiled as host C++ behind the CUDA stand-in of
make_golden_filters.py, launched as cuburn/output.py:21-26 does (32 x 8 blocks over width x height, gutter 12), for
tests/test_cpu_golden.py::test_output_conversion_matches_reference_kernels.

The reference hands every thread BLOCK one ring-buffer slot of 256 RNG states; this design hands state t the pixels t, t + n, ...
(documented in include/flame_hip.h): WHICH dither value a pixel gets differs by design, so the vectors pin everything else — the
clamp, the scale, the YUV matrices, studio swing, the un-dithered Cb plane of 4:4:4 10-bit, the alpha-weighted 4:2:0 chroma, the
plane layouts — to within the dither's one code value, and exactly where no dither applies (black, saturated).
    python tests/golden/make_golden_output.py          (in the build container: needs /root/reference and g++)
"""
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

PRE = MF.PRELUDE.replace('extern "C" {', r'''
#include <algorithm>
using std::max; using std::min;
struct uchar3 { unsigned char x, y, z; }; struct uchar4 { unsigned char x, y, z, w; };
struct ushort3 { unsigned short x, y, z; }; struct ushort4 { unsigned short x, y, z, w; };
static inline uchar3 make_uchar3(float x, float y, float z) { uchar3 r = {(unsigned char)x, (unsigned char)y, (unsigned char)z}; return r; }
static inline uchar4 make_uchar4(float x, float y, float z, float w) { uchar4 r = {(unsigned char)x, (unsigned char)y, (unsigned char)z, (unsigned char)w}; return r; }
static inline ushort3 make_ushort3(float x, float y, float z) { ushort3 r = {(unsigned short)x, (unsigned short)y, (unsigned short)z}; return r; }
static inline ushort4 make_ushort4(float x, float y, float z, float w) { ushort4 r = {(unsigned short)x, (unsigned short)y, (unsigned short)z, (unsigned short)w}; return r; }
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
// a float stored into a uint16_t plane converts as the device does (cvt.rzi.u16.f32 saturates; in host C++ an out-of-range
// conversion is undefined): f32_to_yuv444p10 stores its Cb plane without a clamp
struct cuda_u16 { unsigned short v; cuda_u16() : v(0) {} cuda_u16(float f) : v(f >= 65535.0f ? 65535 : f > 0.0f ? (unsigned short)f : 0) {}
                  cuda_u16(unsigned short u) : v(u) {} operator unsigned short() const { return v; } };
#define uint16_t cuda_u16
extern "C" {''')

MAIN = r'''
}
template <class F> static void launchC(int w, int h, F f) {
    gridDim = {(unsigned)(w + 31) / 32, (unsigned)(h + 7) / 8, 1}; blockDim = {32, 8, 1};
    for (unsigned by = 0; by < gridDim.y; ++by) for (unsigned bx = 0; bx < gridDim.x; ++bx)
        for (unsigned ty = 0; ty < 8; ++ty) for (unsigned tx = 0; tx < 32; ++tx) {          // (thread (0, 0) first: it draws the block's ring-buffer slot)
            blockIdx = {bx, by, 0}; threadIdx = {tx, ty, 0};
            f();
        }
}
int main(int argc, char **argv) {
    FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
    int hdr[4];
    fread(hdr, 4, 4, in);                                  // w, h, astride, aheight
    const int w = hdr[0], h = hdr[1], S = hdr[2];
    std::vector<float4> img((size_t)S * hdr[3]); std::vector<mwc_st> seeds(RB_SIZE_MASK * 256 + 256);
    fread(img.data(), 16, img.size(), in); fread(seeds.data(), 12, seeds.size(), in);
    std::vector<unsigned char> dst((size_t)w * (h + 8) * 16);
#define RUN(K, T, BYTES) { std::vector<mwc_st> rng = seeds; ringbuf rb = {0, 0}; memset(dst.data(), 0, dst.size()); \
        launchC(w, h, [&] { K((T *)dst.data(), img.data(), 12, w, S, h, &rb, rng.data()); }); fwrite(dst.data(), 1, (BYTES), out); }
    const size_t n = (size_t)w * h;
    RUN(f32_to_rgba_u8, uchar4, 4 * n)
    RUN(f32_to_rgba_u16, ushort4, 8 * n)
    RUN(f32_to_yuv444p, char, 3 * n)
    RUN(f32_to_yuv444p10, uint16_t, 6 * n)
    RUN(f32_to_yuv420p10, uint16_t, 3 * n)
    RUN(f32_to_yuv444p12, uint16_t, 6 * n)
    fclose(out);
    return 0;
}
'''


def main():
    tmp, dst = MG.prepare_reference()
    from cuburn.code import output as co, util, mwc as ref_mwc
    from cuburn import render
    src = util.assemble_code(co.pixfmtlib).replace('#include<cuda.h>', '')
    src, n2 = re.subn(r'asm\("cvt\.rni\.s32\.f32\s+%0,\s+%1;" : "=r"\(ret\) : "f"\(f\)\);', 'ret = (uint32_t)(int32_t)rintf(f);', src)
    assert n2 == 1 and ref_mwc.mwclib.defs in src
    src = src.replace(ref_mwc.mwclib.defs, r'''
static uint32_t mwc_next(mwc_st &st) { uint64_t t = (uint64_t)st.mul * st.state + st.carry; st.state = (uint32_t)t; st.carry = (uint32_t)(t >> 32); return st.state; }
static float mwc_next_01(mwc_st &st) { return mwc_next(st) * (1.0f / 4294967296.0f); }
static float mwc_next_11(mwc_st &st) { return (float)(int32_t)mwc_next(st) * (1.0f / 2147483648.0f); }
''')
    main_src = MAIN
    work = tempfile.mkdtemp(prefix='output_ref_')
    open(os.path.join(work, 'k.cpp'), 'w').write(PRE + src + main_src)
    r = subprocess.run(['g++', '-O1', '-std=c++17', '-ffp-contract=off', '-fno-fast-math', '-w', '-o', os.path.join(work, 'k'), os.path.join(work, 'k.cpp')],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:4000]
    w, h = 64, 24                                              # whole blocks (the kernels' `x > dstride` test lets x == dstride through)
    d = render.Framebuffers.calc_dim(w, h)
    S, AH = int(d.astride), int(d.ah)
    rs = np.random.RandomState(31)
    img = rs.uniform(-0.15, 1.2, (AH, S, 4)).astype(np.float32)          # below black, in range, above white
    img[rs.uniform(size=(AH, S)) < 0.1] = 0.0
    img[..., 3] = np.clip(img[..., 3], 0, None)                            # (alpha weights the 4:2:0 chroma)
    rb_size = int(re.search(r'#define RB_SIZE_MASK (\d+)', src).group(1)) + 1
    seeds = ref_mwc.make_seeds(rb_size * 256, host_seed=77)
    with open(os.path.join(work, 'in.bin'), 'wb') as fp:
        fp.write(np.array([w, h, S, AH], np.int32).tobytes()); fp.write(img.tobytes()); fp.write(np.ascontiguousarray(seeds, np.uint32).tobytes())
    subprocess.run([os.path.join(work, 'k'), os.path.join(work, 'in.bin'), os.path.join(work, 'out.bin')], check=True)
    raw = open(os.path.join(work, 'out.bin'), 'rb').read()
    n = w * h
    out = {'width': np.int32(w), 'height': np.int32(h), 'image': img}
    at = 0
    for name, dt, count in (('rgba_u8', np.uint8, 4 * n), ('rgba_u16', np.uint16, 4 * n), ('yuv444p', np.uint8, 3 * n),
                            ('yuv444p10', np.uint16, 3 * n), ('yuv420p10', np.uint16, 3 * n // 2), ('yuv444p12', np.uint16, 3 * n)):
        nb = count * np.dtype(dt).itemsize
        out['out_' + name] = np.frombuffer(raw[at:at + nb], dt).copy(); at += nb
    assert at == len(raw), (at, len(raw))
    np.savez_compressed(os.path.join(HERE, 'output_formats.npz'), **out)
    print('wrote output_formats.npz')


if __name__ == '__main__':
    main()
