#!/usr/bin/env python3
"""Golden vectors for the filter kernels — density blurs, the bilateral (density-estimation) filter, log scale, the clip / tone-map
kernels, YUV -> RGB, log encode — produced by the REFERENCE's own CUDA text, for
tests/test_cpu_golden.py::test_filter_kernels_match_reference_kernels.

cuburn/code/filters.py and cuburn/code/color.py hold the kernels as CUDA source strings (no templates); the reference compiles them
with nvcc at run time and reads its images through textures.  Here the assembled module text (cuburn/code/util.py:83-93
``assemble_code``, taken from the imported reference at generation time and kept nowhere) is compiled as HOST C++ behind a
forty-line CUDA stand-in: ``__global__`` / ``__device__`` / ``__constant__`` / ``__shared__`` as storage classes, float2/3/4,
threadIdx / blockIdx / blockDim / gridDim as globals set by a loop over the reference's launch shape (cuburn/code/util.py:45-53:
32 x 8 blocks over astride x aheight), ``texture<T, 2>`` + ``tex2D`` as a point fetch with clamped addressing (what the hardware
does for unnormalised coordinates whatever mode is asked for, cuburn/code/util.py:55-60), the one inline-PTX statement of
``tex_shear`` (cvt.rni: round to nearest even) as rintf.  the ``__shared__`` table of ``bilateral`` is filled by the first row of a block's threads:
that launch runs twice, the second pass sees the complete table (its output buffer is not its input).  Host-side launch sequences follow cuburn/filters.py
(Bilateral.apply: 60-95, HaloClip: 111-128, SmearClip: 138-160).

Kept: the input image, the scalar arguments, and every kernel's output AT 640 SAMPLED POSITIONS (filters.npz, ~250 KB).
    python tests/golden/make_golden_filters.py          (in the build container: needs /root/reference and g++)
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

PRELUDE = r'''
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#define __global__
#define __device__
#define __constant__ static
#define __shared__ static
#define __syncthreads() ((void)0)
#define __noinline__
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
static inline float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
struct idx3 { unsigned x, y, z; };
static idx3 threadIdx, blockIdx, blockDim, gridDim;
enum { cudaTextureType2D = 2 };
template <typename T, int D> struct texture { const T *data; int w, h; };
template <typename T> static inline T tex2D(const texture<T, 2> &t, float x, float y) {
    int ix = (int)floorf(x), iy = (int)floorf(y);
    ix = ix < 0 ? 0 : ix >= t.w ? t.w - 1 : ix;
    iy = iy < 0 ? 0 : iy >= t.h ? t.h - 1 : iy;
    return t.data[(size_t)iy * t.w + ix];
}
extern "C" {
'''

MAIN = r'''
}
template <class F> static void launch(int astride, int ah, F f, int passes = 1) {
    gridDim = {(unsigned)astride / 32, (unsigned)ah / 8, 1}; blockDim = {32, 8, 1};
    for (int pass = 0; pass < passes; ++pass)                 // (bilateral's __shared__ table: complete on a second pass; its output is not its input)
        for (unsigned by = 0; by < gridDim.y; ++by) for (unsigned bx = 0; bx < gridDim.x; ++bx)
            for (unsigned ty = 0; ty < 8; ++ty) for (unsigned tx = 0; tx < 32; ++tx) {
                blockIdx = {bx, by, 0}; threadIdx = {tx, ty, 0};
                f();
            }
}
static FILE *out;
static void put(const void *p, size_t bytes) { fwrite(p, 1, bytes, out); }
int main(int argc, char **argv) {
    FILE *in = fopen(argv[1], "rb"); out = fopen(argv[2], "wb");
    int W, H; float a[16], k1[7], kw[7];
    fread(&W, 4, 1, in); fread(&H, 4, 1, in); fread(a, 4, 16, in); fread(k1, 4, 7, in); fread(kw, 4, 7, in);
    const size_t n = (size_t)W * H;
    std::vector<float4> img(n), front(n), back(n), left(n), tmp4(n);
    std::vector<float> b1(n), b2(n);
    fread(img.data(), 16, n, in);
    auto coefs = [&](const float *k) { memcpy(gauss_coefs, k, 28); };
    auto tex4 = [&](const float4 *p) { chan4_src.data = p; chan4_src.w = W; chan4_src.h = H; };
    auto tex1 = [&](const float *p) { chan1_src.data = p; chan1_src.w = W; chan1_src.h = H; };
    // a: 0 k1, 1 k2, 2 degamma, 3 sstd, 4 cstd, 5 dstd, 6 dpow, 7 gspeed, 8 gam (1 / gamma), 9 lin, 10 lingam, 11 vib, 12 highpow, 13 brightness, 14 halo gamma - 1
    coefs(k1);
    launch(W, H, [&] { logscale(tmp4.data(), img.data(), a[0], a[1]); }); put(tmp4.data(), 16 * n);
    launch(W, H, [&] { yuv_to_rgb(tmp4.data(), img.data()); }); put(tmp4.data(), 16 * n);
    launch(W, H, [&] { logencode(tmp4.data(), img.data(), a[2]); }); put(tmp4.data(), 16 * n);
    for (int p = 0; p < 8; ++p) {                                   // one direction at a time, each from the same image
        tex4(img.data());
        launch(W, H, [&] { den_blur(b1.data(), p, 0); }); put(b1.data(), 4 * n);
        tex1(b1.data());
        launch(W, H, [&] { den_blur_1c(b2.data(), p, 1); }); put(b2.data(), 4 * n);
        tex1(b2.data());
        launch(W, H, [&] { bilateral(tmp4.data(), p, 15, a[3], a[4], a[5], a[6], a[7]); }, 2); put(tmp4.data(), 16 * n);
    }
    for (int p = 2; p < 4; ++p) { tex4(img.data()); launch(W, H, [&] { full_blur(tmp4.data(), p, 0); }); put(tmp4.data(), 16 * n); }
    {   // Bilateral.apply: eight directions, front and back flipped after each
        front = img;
        float4 *f = front.data(), *b = back.data();
        for (int p = 0; p < 8; ++p) {
            tex4(f);
            launch(W, H, [&] { den_blur((float *)b, p, 0); });
            tex1((float *)b);
            launch(W, H, [&] { den_blur_1c((float *)left.data(), p, 1); });
            tex1((float *)left.data());
            launch(W, H, [&] { bilateral(b, p, 15, a[3], a[4], a[5], a[6], a[7]); }, 2);
            std::swap(f, b);
        }
        put(f, 16 * n);
    }
    {   // HaloClip.apply
        front = img;
        launch(W, H, [&] { apply_gamma((float *)left.data(), front.data(), 0.1f); });
        tex1((float *)left.data()); launch(W, H, [&] { den_blur_1c((float *)back.data(), 2, 0); });
        tex1((float *)back.data()); launch(W, H, [&] { den_blur_1c((float *)left.data(), 3, 0); });
        launch(W, H, [&] { haloclip(front.data(), (float *)left.data(), a[14]); });
        put(front.data(), 16 * n);
    }
    {   // SmearClip.apply (blur width from the profile)
        coefs(kw);
        front = img;
        launch(W, H, [&] { apply_gamma_full_hi(left.data(), front.data(), a[8] - 1.0f); });
        tex4(left.data()); launch(W, H, [&] { full_blur(back.data(), 2, 0); });
        tex4(back.data()); launch(W, H, [&] { full_blur(left.data(), 3, 0); });
        tex4(left.data()); launch(W, H, [&] { full_blur(back.data(), 0, 0); });
        tex4(back.data()); launch(W, H, [&] { full_blur(left.data(), 1, 0); });
        launch(W, H, [&] { smearclip(front.data(), left.data(), a[8] - 1.0f, a[9], a[10]); });
        put(front.data(), 16 * n);
        coefs(k1);
    }
    front = img; launch(W, H, [&] { plainclip(front.data(), a[8] - 1.0f, a[9], a[10], a[13]); }); put(front.data(), 16 * n);
    front = img; launch(W, H, [&] { colorclip(front.data(), a[11], a[12], a[8], a[9], a[10]); }); put(front.data(), 16 * n);
    fclose(out);
    return 0;
}
'''


def main():
    tmp, dst = MG.prepare_reference()
    from cuburn.code import filters as cf, util
    from cuburn import render
    src = util.assemble_code(cf.logscalelib, cf.yuvfilterlib, cf.logencodelib, cf.fullblurlib, cf.bilaterallib,
                             cf.halocliplib, cf.smearcliplib, cf.plaincliplib, cf.colorcliplib)
    src = src.replace('#include<cuda.h>', '')
    src, n1 = re.subn(r'asm\("\{\\n\\t"\s*"cvt\.rni\.ftz\.f32\.f32\s+%0, %0;\\n\\t"\s*"cvt\.rni\.ftz\.f32\.f32\s+%1, %1;\\n\\t"\s*"\}\\n" : "\+f"\(i\), "\+f"\(j\)\);',
                      'i = rintf(i); j = rintf(j);', src)
    src, n2 = re.subn(r'asm\("cvt\.rni\.s32\.f32\s+%0,\s+%1;" : "=r"\(ret\) : "f"\(f\)\);', 'ret = (uint32_t)(int32_t)rintf(f);', src)
    assert n1 == 1 and n2 == 1, (n1, n2)
    work = tempfile.mkdtemp(prefix='filters_ref_')
    open(os.path.join(work, 'k.cpp'), 'w').write(PRELUDE + src + MAIN)
    r = subprocess.run(['g++', '-O1', '-ffp-contract=off', '-fno-fast-math', '-w', '-o', os.path.join(work, 'k'), os.path.join(work, 'k.cpp')],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:4000]

    w, h = 36, 20
    d = render.Framebuffers.calc_dim(w, h)
    W, H = int(d.astride), int(d.ah)
    rs = np.random.RandomState(77)
    n = W * H
    # an accumulator as the flush leaves it: density in .w (a fifth of the cells empty, the rest 0.01 .. 3000 hits, a smooth blob under
    # the noise so that the gradient factor sees structure), colour sums = density * (y, u + 0.5, v + 0.5)
    yy, xx = np.mgrid[0:H, 0:W]
    blob = np.exp(-((xx - 0.55 * W) ** 2 + (yy - 0.45 * H) ** 2) / (2 * 9.0 ** 2))
    den = (10 ** rs.uniform(-2, 1.2, (H, W)) * (0.05 + 40 * blob)).astype(np.float32)
    den[rs.uniform(size=(H, W)) < 0.2] = 0
    col = rs.uniform(0, 1, (H, W, 3)).astype(np.float32)
    img = np.zeros((H, W, 4), np.float32)
    img[..., :3] = col * den[..., None]
    img[..., 3] = den
    gamma, lin = 2.5, 0.03
    gam = np.float32(1 / gamma)
    lingam = np.float32(np.float32(lin) ** (gam - 1.0))
    args = np.array([1.3 * 268 / 256, 1 / 37.0, 2.2, 6.0 * w / 1920. * 40, 0.05, 0.25, 0.8, 1.5,
                     gam, lin, lingam, 0.8, 0.6, 1.1, 1 / gamma - 1, 0], np.float32)
    k1 = np.exp(np.float32(np.arange(-3, 4)) ** 2 / (-2 * 1 ** 2)).astype(np.float32); k1 /= np.sum(k1)
    width = 1.7
    kw = np.exp(np.float32(np.arange(-3, 4)) ** 2 / (-2 * width ** 2)).astype(np.float32); kw /= np.sum(kw)
    with open(os.path.join(work, 'in.bin'), 'wb') as fp:
        fp.write(np.array([W, H], np.int32).tobytes()); fp.write(args.tobytes()); fp.write(k1.tobytes()); fp.write(kw.tobytes()); fp.write(img.tobytes())
    subprocess.run([os.path.join(work, 'k'), os.path.join(work, 'in.bin'), os.path.join(work, 'out.bin')], check=True)
    raw = np.fromfile(os.path.join(work, 'out.bin'), np.float32)
    names = ['logscale', 'yuv_to_rgb', 'logencode']
    chans = [4, 4, 4]
    for p in range(8):
        names += ['den_blur_%d' % p, 'den_blur_1c_%d' % p, 'bilateral_%d' % p]; chans += [1, 1, 4]
    names += ['full_blur_2', 'full_blur_3', 'bilateral_chain', 'haloclip_chain', 'smearclip_chain', 'plainclip', 'colorclip']
    chans += [4] * 7
    assert raw.size == n * sum(chans), (raw.size, n * sum(chans))
    pos = np.sort(rs.choice(n, 640, replace=False)).astype(np.int32)
    out = {'width': np.int32(w), 'height': np.int32(h), 'astride': np.int32(W), 'aheight': np.int32(H), 'image': img, 'args': args,
           'smear_width': np.float32(width), 'positions': pos}
    at = 0
    for nm, c in zip(names, chans):
        a = raw[at:at + n * c].reshape(n, c); at += n * c
        out['out_' + nm] = a[pos].copy()
    np.savez_compressed(os.path.join(HERE, 'filters.npz'), **out)
    print('wrote filters.npz:', len(names), 'kernel outputs at', len(pos), 'positions of a', W, 'x', H, 'accumulator')


if __name__ == '__main__':
    main()
