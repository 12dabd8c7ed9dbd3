#!/usr/bin/env python3
"""A histogram computed by the REFERENCE's own iterate kernel, for tests/test_cpu_golden.py::test_histogram_of_reference_iterate_kernel.

``cuburn.code.iter.mkiterlib`` renders the CUDA text of the whole iterate module for a genome: the interpolation kernel, one
``apply_xf`` per xform and ``iter`` itself — 256 threads a block that walk 256 rounds, choose their xform per warp from a shared
``cosel[]``, exchange points through shared memory behind a ``__syncthreads()`` every round and add their samples to the packed
histogram (cuburn/code/iter.py:157-418).  Here that text runs on the host, unmodified but for what a host compiler cannot take:
  * a block's 256 threads are 256 coroutines (ucontext) of ONE OS thread; ``__syncthreads()`` yields to a scheduler that resumes
    thread 0 .. 255 in turn, so every thread finishes a phase before any starts the next (barrier semantics), deterministically;
  * the inline PTX of the sample's add (iter.py:332-411) is replaced by a C restatement — the one piece of this harness that is not
    the reference's text; it is held to that PTX, bit for bit, by make_golden_ptx.py / test_packed_cell_add_and_flush_match_reference_ptx;
    flush_atom (all PTX) is not run: the test flushes with the oracle's flush, pinned the same way;
  * mwc_next / _01 / _11 (inline PTX) by the three-line restatement pinned by mwc.json; ``trunca``'s cvt.rni by rintf;
  * the reference's launch shape (cuburn/render.py:343-346): grid (1024 temporal samples, 3), blocks of 32 x 8, a ring buffer of 1024
    chunks of points / RNG states; every chunk is used three times, its first use is the un-plotted "fuse" pass (iter.py:207-215).
The parameter blocks come from the same module's ``interp_iter_params`` run first; the packed palette from the oracle's
``interp_palette`` (pinned bit-exactly to the reference's kernel by interp_palette.npz).
Kept (iter_hist.npz), for cfg2, cfg3 and cfg5 at 320 x 180: 8 x 8-block sums of the flushed accumulator (density and colour sums), the
number of samples attempted and plotted.
    python tests/golden/make_golden_iter.py          (in the build container: needs /root/reference and g++; about five minutes)
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
sys.path.insert(0, os.path.join(REPO, 'tests'))
import make_golden as MG          # noqa: E402
import make_golden_filters as MF  # noqa: E402

NTS, ROWS = 1024, 3

PRE = MF.PRELUDE.replace('#define __syncthreads() ((void)0)', 'static void co_yield_();\n#define __syncthreads() co_yield_()').replace('extern "C" {', r'''
#include <algorithm>
#include <ucontext.h>
using std::max; using std::min; using std::isfinite;
struct uint2 { uint32_t x, y; };
enum { cudaSurfaceType2D = 2 };
template <typename T, int D> struct surface { const uint64_t *data; };
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static uint64_t g_plotted = 0;
static const uint64_t *g_pal;            // [64][256] packed cells
// C restatement of the inline PTX of cuburn/code/iter.py:332-411 (held to it by make_golden_ptx.py)
static void ptx_accum(float cc, float dither, int time, uint32_t i, uint64_t atom_ptr, float cosel, uint64_t out_ptr, float mult) {
    float colorf = fmaf(cc, 255.0f, dither);
    uint32_t color = colorf != colorf || colorf <= 0.0f ? 0u : colorf >= 4294967295.0f ? 0xffffffffu : (uint32_t)rintf(colorf);
    const uint64_t val = g_pal[(size_t)std::min(std::max(time, 0), 63) * 256 + std::min(color, 255u)];
    uint64_t *cell = (uint64_t *)atom_ptr + i;
    ++g_plotted;
    if (cosel <= 0.97f) { *cell += val; return; }
    const uint64_t old = *cell; *cell = old + val;
    if ((uint32_t)(old >> 32) < (256u << 23)) return;
    const uint64_t cur = *cell; *cell = 0;
    const uint32_t hi = (uint32_t)(cur >> 32), lo = (uint32_t)cur;
    if (hi == 0) return;
    const float d = (float)(hi >> 22), y = (float)((hi >> 4) & 0x3ffff), u = (float)(((hi & 0xf) << 14) | (lo >> 18)), v = (float)(lo & 0x3ffff);
    const float m = mult * (float)(1.0 / 255.0);
    float *o = (float *)out_ptr + 4 * (size_t)i;
    o[0] += y * m; o[1] += u * m; o[2] += v * m; o[3] += d * mult;
}
extern "C" {''')

MAIN = r'''
}
// ---- a block's threads as coroutines of one OS thread
static ucontext_t main_ctx, ctx[256];
static bool done[256];
static int cur;
static void co_yield_() { swapcontext(&ctx[cur], &main_ctx); }
static uint64_t a_out, a_atom; static ringbuf *a_rb; static mwc_st *a_msts; static float4 *a_points; static const uint32_t *a_hot; static const iter_params *a_params;
static void entry() { iter(a_out, a_atom, a_rb, a_msts, a_points, a_hot, a_params); done[cur] = true; swapcontext(&ctx[cur], &main_ctx); }
static std::vector<char> stacks;
static void run_block(unsigned bx, unsigned by) {
    blockIdx = {bx, by, 0};
    for (int t = 0; t < 256; ++t) {
        getcontext(&ctx[t]);
        ctx[t].uc_stack.ss_sp = &stacks[(size_t)t * 65536]; ctx[t].uc_stack.ss_size = 65536; ctx[t].uc_link = &main_ctx;
        makecontext(&ctx[t], entry, 0);
        done[t] = false;
    }
    for (bool any = true; any;) {
        any = false;
        for (int t = 0; t < 256; ++t) if (!done[t]) {
            cur = t; threadIdx = {(unsigned)t % 32, (unsigned)t / 32, 0};
            swapcontext(&main_ctx, &ctx[t]);
            any = true;
        }
    }
}
int main(int argc, char **argv) {
    FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
    int hdr[9]; float tt[2];
    fread(hdr, 4, 9, in); fread(tt, 4, 2, in);     // knot rows, row length, temporal samples, grid rows, width, height, awidth, aheight, astride; tstart, tstep
    const size_t nk = (size_t)hdr[0] * hdr[1];
    std::vector<float> times(nk), knots(nk);
    fread(times.data(), 4, nk, in); fread(knots.data(), 4, nk, in);
    std::vector<uint64_t> pal(64 * 256); fread(pal.data(), 8, pal.size(), in); g_pal = pal.data();
    const int nts = hdr[2];
    std::vector<mwc_st> msts((size_t)nts * 256); fread(msts.data(), 12, msts.size(), in);
    acc_size.width = hdr[4]; acc_size.height = hdr[5]; acc_size.awidth = hdr[6]; acc_size.aheight = hdr[7]; acc_size.astride = hdr[8];
    const size_t ncell = (size_t)hdr[7] * hdr[8];
    stacks.resize((size_t)256 * 65536);
    // parameter blocks: the module's own interpolation kernel, one thread per temporal sample
    std::vector<iter_params> P(nts);
    blockDim = {256, 1, 1}; gridDim = {1, 1, 1}; blockIdx = {0, 0, 0};
    for (int id = 0; id < nts; ++id) { threadIdx = {(unsigned)id, 0, 0}; interp_iter_params(P.data(), times.data(), knots.data(), tt[0], tt[1], nts); }
    std::vector<uint64_t> atom(ncell, 0); std::vector<float4> acc(ncell, make_float4(0, 0, 0, 0));
    std::vector<float4> points((size_t)nts * 256, make_float4(NAN, NAN, NAN, NAN));       // cuburn/render.py:327: filled with NaN
    std::vector<uint32_t> hot(ncell / 2, 0);
    ringbuf rb = {0, 0};
    a_out = (uint64_t)acc.data(); a_atom = (uint64_t)atom.data(); a_rb = &rb; a_msts = msts.data(); a_points = points.data(); a_hot = hot.data(); a_params = P.data();
    blockDim = {32, 8, 1}; gridDim = {(unsigned)nts, (unsigned)hdr[3], 1};
    for (int by = 0; by < hdr[3]; ++by) for (int bx = 0; bx < nts; ++bx) run_block(bx, by);
    fwrite(&g_plotted, 8, 1, out); fwrite(atom.data(), 8, ncell, out); fwrite(acc.data(), 16, ncell, out);
    fclose(out);
    return 0;
}
'''


def main():
    tmp, dst = MG.prepare_reference()
    from cuburn.code import iter as ref_iter, util, mwc as ref_mwc
    from cuburn import render
    from cuburn_amd import configs
    from common import O, oracle_palette
    out = {}
    for name in ('cfg2', 'cfg3', 'cfg5'):
        gnm, prof = configs.CONFIGS[name]()
        prof = dict(prof, width=320, height=180)
        run_genome(name, gnm, prof, ref_iter, util, ref_mwc, render, O, oracle_palette, out)
    np.savez_compressed(os.path.join(HERE, 'iter_hist.npz'), **out)
    print('wrote iter_hist.npz')


def run_genome(name, gnm, prof, ref_iter, util, ref_mwc, render, O, oracle_palette, out):
    from cuburn_amd import profile as my_profile
    from common import frame_times
    packer, lib = ref_iter.mkiterlib(gnm)
    src = util.assemble_code(lib).replace('#include<cuda.h>', '')
    src, n = re.subn(r'asm\("cvt\.rni\.s32\.f32\s+%0,\s+%1;" : "=r"\(ret\) : "f"\(f\)\);', 'ret = (uint32_t)(int32_t)rintf(f);', src)
    assert n == 1 and ref_mwc.mwclib.defs in src
    src = src.replace(ref_mwc.mwclib.defs, r'''
static uint32_t mwc_next(mwc_st &st) { uint64_t t = (uint64_t)st.mul * st.state + st.carry; st.state = (uint32_t)t; st.carry = (uint32_t)(t >> 32); return st.state; }
static float mwc_next_01(mwc_st &st) { return mwc_next(st) * (1.0f / 4294967296.0f); }
static float mwc_next_11(mwc_st &st) { return (float)(int32_t)mwc_next(st) * (1.0f / 2147483648.0f); }
''')
    blocks = list(re.finditer(r'asm volatile \(("(?:[^"\\]|\\.)*")\s*::(.*?)\);', src, re.S))
    assert len(blocks) == 2
    args = re.findall(r'"\w"\((.*?)\)(?:,|$)', ' '.join(blocks[0].group(2).split()))
    assert len(args) == 8, args
    src = src[:blocks[0].start()] + 'ptx_accum(%s);' % ', '.join(args) + src[blocks[0].end():blocks[1].start()] + ';' + src[blocks[1].end():]
    src, n = re.subn(r'asm\("trap;"\);', 'abort();', src)
    src, n2 = re.subn(r'#define RB_SIZE_MASK \d+', '#define RB_SIZE_MASK %d' % (NTS - 1), src)
    assert n2 == 1
    work = tempfile.mkdtemp(prefix='iter_ref_%s_' % name)
    open(os.path.join(work, 'k.cpp'), 'w').write(PRE + src + MAIN)
    r = subprocess.run(['g++', '-O1', '-ffp-contract=off', '-fno-fast-math', '-w', '-o', os.path.join(work, 'k'), os.path.join(work, 'k.cpp')],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:5000]
    times, knots = packer.pack(gnm)
    times, knots = np.ascontiguousarray(times, np.float32), np.ascontiguousarray(knots, np.float32)
    d = render.Framebuffers.calc_dim(prof['width'], prof['height'])
    ts, td = [np.float32(v) for v in frame_times(my_profile.wrap(prof, gnm), 0.5)]       # (cuburn/render.py:410-411)
    seeds = ref_mwc.make_seeds(NTS * 256, host_seed=4242)
    pal_seeds = ref_mwc.make_seeds(64 * 256, host_seed=4243)
    pal, _ = oracle_palette(gnm, ts, td, pal_seeds)
    with open(os.path.join(work, 'in.bin'), 'wb') as fp:
        fp.write(np.array([times.shape[0], times.shape[1], NTS, ROWS, d.w, d.h, d.aw, d.ah, d.astride], np.int32).tobytes())
        fp.write(np.array([ts, td / NTS], np.float32).tobytes())
        fp.write(times.tobytes()); fp.write(knots.tobytes())
        fp.write(np.ascontiguousarray(pal, np.uint64).tobytes()); fp.write(np.ascontiguousarray(seeds, np.uint32).tobytes())
    subprocess.run([os.path.join(work, 'k'), os.path.join(work, 'in.bin'), os.path.join(work, 'out.bin')], check=True)
    raw = open(os.path.join(work, 'out.bin'), 'rb').read()
    ncell = int(d.ah) * int(d.astride)
    plotted = int(np.frombuffer(raw[:8], np.uint64)[0])
    atom = np.frombuffer(raw[8:8 + 8 * ncell], np.uint64).copy()
    acc = np.frombuffer(raw[8 + 8 * ncell:], np.float32).reshape(ncell, 4).copy()
    dim = O.calc_dim(prof['width'], prof['height'])
    hot = np.zeros((ncell + 15) // 16, np.uint32)
    O.flush(dim, atom, acc, hot)
    attempted = NTS * (ROWS - 1) * 65536                     # every chunk's first use is the fuse pass
    H, W = int(d.ah) // 8 * 8, int(d.astride) // 8 * 8
    img = acc.reshape(int(d.ah), int(d.astride), 4).astype(np.float64)
    blocks8 = img[:H, :W].reshape(H // 8, 8, W // 8, 8, 4).sum((1, 3)).astype(np.float32)
    out[name + '_size'] = np.array([prof['width'], prof['height']], np.int32)
    out[name + '_blocks8'] = blocks8
    out[name + '_counts'] = np.array([attempted, plotted, int(round(img[..., 3].sum()))], np.int64)
    print(name, 'plotted', plotted, 'of', attempted, '= %.4f;' % (plotted / attempted), 'density in accumulator %.0f' % img[..., 3].sum())


if __name__ == '__main__':
    main()
