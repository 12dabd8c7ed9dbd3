// filters.hip — density-estimation (directional bilateral), log-scale and tone-map kernels.
//
// Device side of cuburn/code/filters.py (all kernels) for gfx950.  Images are the padded
// accumulation buffers (ah rows x astride float4 / float).  The reference reads through
// CUDA textures (POINT filter, unnormalised coordinates => edge clamp); CDNA has no texture
// path worth using for this, so taps are explicit clamped loads.
#include "flame_device.h"
#include "kernels.h"
#include "tone_device.h"
#include <utility>
#include <type_traits>
#include <cstdlib>

// cuburn/code/filters.py:8-17
__constant__ float2 shear_patterns[16] = {
    {1.0f, 0.0f}, {0.0f, 1.0f}, {1.0f, 1.0f}, {-1.0f, 1.0f},
    {1.0f, 0.5f}, {-0.5f, 1.0f}, {1.0f, -0.5f}, {0.5f, 1.0f},
    {1.0f, 0.666667f}, {-0.666667f, 1.0f}, {1.0f, -0.666667f}, {0.666667f, 1.0f},
    {1.0f, 0.333333f}, {-0.333333f, 1.0f}, {1.0f, -0.333333f}, {0.333333f, 1.0f},
};

struct Coefs7 { float c[7]; };

#define PIX_IDX(d)                                                            \
    const int xi = blockIdx.x * blockDim.x + threadIdx.x;                     \
    const int yi = blockIdx.y * blockDim.y + threadIdx.y;                     \
    const int gi = yi * (int)(d).astride + xi

// cuburn/code/filters.py:22-35 tex_shear: offset rounded to nearest-even before adding x, y;
// clamped addressing.
__device__ __forceinline__ int shear_idx(const fl_dim &d, float2 pat, int x, int y, float radius) {
    int i = (int)__builtin_rintf(pat.x * radius), j = (int)__builtin_rintf(pat.y * radius);
    int xs = min(max(x + i, 0), (int)d.astride - 1), ys = min(max(y + j, 0), (int)d.ah - 1);
    return ys * (int)d.astride + xs;
}

// cuburn/code/filters.py:71-77 + cuburn/code/color.py:25-40
__device__ __forceinline__ float4 yuv_px(float4 p) {
    float u = p.y - 0.5f * p.w, v = p.z - 0.5f * p.w;
    float4 o;
    o.x = fmaxf(0.0f, p.x + 1.402f * v);
    o.y = fmaxf(0.0f, p.x - 0.34414f * u - 0.71414f * v);
    o.z = fmaxf(0.0f, p.x + 1.772f * u);
    o.w = p.w;
    return o;
}
__global__ void __launch_bounds__(256) k_yuv_to_rgb(fl_dim d, float4 *__restrict__ dst, const float4 *__restrict__ src) {
    PIX_IDX(d);
    dst[gi] = yuv_px(src[gi]);
}

// cuburn/code/filters.py:106-117
__global__ void __launch_bounds__(256)
k_den_blur(fl_dim d, float *__restrict__ dst, const float4 *__restrict__ src, int pattern, int upsample, Coefs7 k) {
    PIX_IDX(d);
    const float2 pat = shear_patterns[pattern];
    float den = 0.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i)
        den += src[shear_idx(d, pat, xi, yi, (float)((i - 3) * (1 << upsample)))].w * k.c[i];
    dst[gi] = den;
}

// cuburn/code/filters.py:120-131
__global__ void __launch_bounds__(256)
k_den_blur_1c(fl_dim d, float *__restrict__ dst, const float *__restrict__ src, int pattern, int upsample, Coefs7 k) {
    PIX_IDX(d);
    const float2 pat = shear_patterns[pattern];
    float den = 0.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i)
        den += src[shear_idx(d, pat, xi, yi, (float)((i - 3) * (1 << upsample)))] * k.c[i];
    dst[gi] = den;
}

// cuburn/code/filters.py:136-151
__global__ void __launch_bounds__(256)
k_full_blur(fl_dim d, float4 *__restrict__ dst, const float4 *__restrict__ src, int pattern, int upsample, Coefs7 k) {
    PIX_IDX(d);
    const float2 pat = shear_patterns[pattern];
    float4 v = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float4 p = src[shear_idx(d, pat, xi, yi, (float)((i - 3) * (1 << upsample)))];
        v.x += p.x * k.c[i]; v.y += p.y * k.c[i]; v.z += p.z * k.c[i]; v.w += p.w * k.c[i];
    }
    dst[gi] = v;
}

// cuburn/code/filters.py:166-264
__global__ void __launch_bounds__(256)
k_bilateral(fl_dim d, float4 *__restrict__ dst, const float4 *__restrict__ src, const float *__restrict__ blur,
            int pattern, int radius, float sstd, float cstd, float dstd, float dpow, float gspeed)
{
    PIX_IDX(d);
    __shared__ float spa[32];
    const int lt = threadIdx.y * blockDim.x + threadIdx.x;
    if (lt < 32) { float df = (float)lt; spa[lt] = fexp(fdiv(df * df, -FM_SQRT2 * sstd)); }
    const float2 pat = shear_patterns[pattern];
    const float cscale = frcp(-FM_SQRT2 * 3.0f * cstd);
    const float dscale = fdiv(-0.5f, dstd);

    float4 cen = src[gi];
    const float cdrcp = frcp(cen.w + 1.0e-6f);
    cen.x *= cdrcp; cen.y *= cdrcp; cen.z *= cdrcp;
    const float cpowden = de_pow(cen.w, dpow);

    float4 out = make_float4(0, 0, 0, 0);
    float weightsum = 0.0f;
    __syncthreads();

    float4 pix = src[shear_idx(d, pat, xi, yi, (float)(-radius) - 1.0f)];
    float4 next = src[shear_idx(d, pat, xi, yi, (float)(-radius))];
    for (int r = -radius; r <= radius; ++r) {
        const float prev = pix.w;
        pix = next;
        next = src[shear_idx(d, pat, xi, yi, (float)r + 1.0f)];

        float cdiff = 0.5f;
        if (pix.w > 0.0f && cen.w > 0.0f) {
            const float pdrcp = frcp(pix.w);
            const float yd = pix.x * pdrcp - cen.x, ud = pix.y * pdrcp - cen.y, vd = pix.z * pdrcp - cen.z;
            cdiff = yd * yd + ud * ud + vd * vd;
        }
        const float powden = de_pow(pix.w, dpow);
        const float dfact = fexp2(dscale * fabsf(cpowden - powden));
        const float avg = blur[shear_idx(d, pat, xi, yi, (float)r)];
        float gradfact = fdiv(next.w - prev, avg + 1.0e-6f);
        if (r < 0) gradfact = -gradfact;
        gradfact = fexp2(-fexp2(gspeed * gradfact));
        float factor = spa[abs(r)] * fexp(cscale * cdiff) * dfact;
        if (r != 0) factor *= gradfact;
        weightsum += factor;
        out.x += factor * pix.x; out.y += factor * pix.y; out.z += factor * pix.z; out.w += factor * pix.w;
    }
    const float wr = frcp(weightsum + 1e-10f);
    out.x *= wr; out.y *= wr; out.z *= wr; out.w *= wr;
    dst[gi] = out;
}

// ---------------------------------------------------------------------------------------------
// Restructured DE pass (same arithmetic as k_bilateral, cuburn/code/filters.py:166-264, with the
// tap-invariant work hoisted out of the 31-tap loop):
//   * the image travels between passes as N = (x/w, y/w, z/w, w): the colour-difference term
//     needs the normalised colour of every tap, and the weighted sums only need n*w;
//   * Pw = w^dpow is computed once per pixel by the pass that produces the pixel;
//   * RA = 1/(avg + 1e-6) is written by the second density blur;
//   * the three exponentials of a tap are one: spa * exp2(cs*cdiff + ds*|dpw| - exp2(gs*grad));
//   * the shear pattern is a template parameter, so tap offsets are immediates.
// 2 transcendentals per tap instead of 8; ~35 VALU per tap instead of ~150.
// cuburn/code/filters.py:8-17,26-34: round-to-nearest-even of slope * radius (compile time)
template <int PATTERN> __host__ __device__ __forceinline__ constexpr int tap_dx(int r) {
    constexpr int num[8] = {2, 0, 2, -2, 2, -1, 2, 1};         // slope_x * 2
    const int v = num[PATTERN] * r;                            // = 2 * slope * r
    return (v % 2 == 0) ? v / 2 : ((v - 1) / 2 % 2 == 0 ? (v - 1) / 2 : (v + 1) / 2);
}
template <int PATTERN> __host__ __device__ __forceinline__ constexpr int tap_dy(int r) {
    constexpr int num[8] = {0, 2, 2, 2, 1, 2, -1, 2};          // slope_y * 2
    const int v = num[PATTERN] * r;
    return (v % 2 == 0) ? v / 2 : ((v - 1) / 2 % 2 == 0 ? (v - 1) / 2 : (v + 1) / 2);
}

// ---------------------------------------------------------------------------------------------
// LDS-tiled DE pass.  A workgroup owns a 32x16 output tile; it stages the tile plus the halo its
// taps can reach (|dx| <= HX, |dy| <= HY, compile-time per direction) of N (float4) and of the
// packed plane PR = (w^dpow, 1/(avg+1e-6)) (float2) into LDS with coalesced, edge-clamped row
// loads, then every thread evaluates its taps from LDS at immediate offsets.  Per direction this
// replaces ~95 L1 gathers per pixel (776 B through the 64 B/clk TA port) by ~3-6 row loads per
// pixel and 33 ds_read_b128 + 31 ds_read_b64 at 256 B/clk, and the image-edge clamp happens
// once, at staging time.
#define DE_TW 32
#define DE_TH 16
#ifndef DE_GROUP
#define DE_GROUP 4
#endif
template <int PATTERN> __host__ __device__ __forceinline__ constexpr int de_hx() {
    int m = 0;
    for (int r = -16; r <= 16; ++r) { int v = tap_dx<PATTERN>(r); v = v < 0 ? -v : v; m = v > m ? v : m; }
    return m;
}
template <int PATTERN> __host__ __device__ __forceinline__ constexpr int de_hy() {
    int m = 0;
    for (int r = -16; r <= 16; ++r) { int v = tap_dy<PATTERN>(r); v = v < 0 ? -v : v; m = v > m ? v : m; }
    return m;
}

// ---------------------------------------------------------------------------------------------
// Packed-math form of the LDS-tiled DE pass.  A thread owns the two pixels (i, y) and (i+16, y)
// of the 32x16 tile and evaluates their taps in lockstep on float2 values,
// so the adds / multiplies / fmas of a tap issue as v_pk_*_f32 (two FP32 operations per lane and
// instruction on CDNA3/4) — the scalar form is VALU-issue bound.  For the two pixels' values to
// arrive as a register pair without moves the tile is staged as six planes (x, y, z, w, w^dpow,
// 1/(avg+1e-6)) interleaved by row ([row][plane][column]): the pair is then two floats of one
// plane 16 apart = one ds_read2_b32 at any tap offset.  The row stride is padded to 16 (mod 64)
// floats: the 16 lanes of a row read 16 consecutive banks, the four rows a wave touches fall on
// the four disjoint bank groups, in both halves of the read — no bank conflicts.
#ifndef DE_PK_GROUP
#define DE_PK_GROUP 2
#endif
#ifndef DE_PK_WAVES
#define DE_PK_WAVES 4
#endif
typedef float f2 __attribute__((ext_vector_type(2)));
typedef f2 f2u __attribute__((aligned(4)));
__device__ __forceinline__ f2 f2fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
// Sheared tiles.  A rectangular tile needs a halo of 16 pixels in x AND y for a diagonal
// direction (64x48 staged pixels for 32x16 outputs: 77 KB, two workgroups per CU).  Here row j of
// a tile (and of its halo) starts S(j) = floor(j * K / 2) pixels further right, K/2 being the
// direction's x step per row (K = 2, -2 for the diagonals, 4 / -4 for slopes +-1/2 along x,
// -1 / 1 for slopes -+1/2 along y, 0 for the axes): a tap then lands in (almost) the same column
// of the staged parallelogram, whose width is 32 + 2 * margin with margin 0 or 1 instead of 16.
// For odd K the residual column offset of a tap depends on the parity p of the output row:
// coff(r, p) = coff(r, 0) + p * delta(r), delta in {-1, 0, 1} — three base addresses.
template <int P> __host__ __device__ __forceinline__ constexpr int de_shear_k() {
    constexpr int k[8] = {0, 0, 2, -2, 4, -1, -4, 1};
    return k[P];
}
template <int P> __host__ __device__ __forceinline__ constexpr int de_shear(int j) { return (j * de_shear_k<P>()) >> 1; }
template <int P> __host__ __device__ __forceinline__ constexpr int de_coff(int r, int p) {
    return tap_dx<P>(r) - (de_shear<P>(p + tap_dy<P>(r)) - de_shear<P>(p));
}
template <int P> __host__ __device__ __forceinline__ constexpr int de_margin() {
    int m = 0;
    for (int r = -16; r <= 16; ++r) for (int p = 0; p < 2; ++p) { int v = de_coff<P>(r, p); v = v < 0 ? -v : v; m = v > m ? v : m; }
    return m;
}
template <int P> __host__ __device__ __forceinline__ constexpr int de_span() {        // |S(15)|: extra width of a band of tiles
    int v = de_shear<P>(DE_TH - 1);
    return v < 0 ? -v : v;
}
template <int PATTERN> __host__ __device__ __forceinline__ constexpr int de_pk_row_stride() {
    int rs = 6 * (DE_TW + 2 * de_margin<PATTERN>());
    while (rs % 64 != 16) ++rs;
    return rs;
}

template <int PATTERN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DE_PK_WAVES, DE_PK_WAVES)))
k_de_bilateral_pk(fl_dim d, float4 *__restrict__ Nout, float2 *__restrict__ PRout, float *__restrict__ Wout,
                  const float4 *__restrict__ N, const float2 *__restrict__ PR,
                  float sstd, float cstd, float dstd, float dpow, float gspeed)
{
    constexpr int HY = de_hy<PATTERN>(), K = de_shear_k<PATTERN>(), M = de_margin<PATTERN>();
    constexpr int LW = DE_TW + 2 * M, LH = DE_TH + 2 * HY;
    constexpr int RS = de_pk_row_stride<PATTERN>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *pl = reinterpret_cast<float *>(smem);                         // [LH][6][LW] (+ row padding)

    const int tid = threadIdx.x;
    // x of column 0 of tile row 0; for K > 0 the band starts S(15) to the left so that its last row reaches x = 0
    const int bx0 = (int)blockIdx.x * DE_TW - (K > 0 ? de_shear<PATTERN>(DE_TH - 1) : 0);
    const int by0 = blockIdx.y * DE_TH;
    constexpr int NIT = (LH * LW + 255) / 256;
    float4 tn[NIT];
    float2 tp[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = min(it * 256 + tid, LH * LW - 1);
        const int ly = idx / LW, lx = idx - ly * LW;
        const int gx = min(max(bx0 + (((ly - HY) * K) >> 1) - M + lx, 0), (int)d.astride - 1);
        const int gy = min(max(by0 + ly - HY, 0), (int)d.ah - 1);
        const uint32_t g = (uint32_t)(gy * (int)d.astride + gx);
        tn[it] = N[g];
        tp[it] = PR[g];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * 256 + tid;
        if (idx < LH * LW) {
            const int ly = idx / LW, lx = idx - ly * LW;
            float *row = pl + ly * RS + lx;
            row[0] = tn[it].x; row[LW] = tn[it].y; row[2 * LW] = tn[it].z; row[3 * LW] = tn[it].w;
            row[4 * LW] = tp[it].x; row[5 * LW] = tp[it].y;
        }
    }
    const float cs2 = frcp(-FM_SQRT2 * 3.0f * cstd) * FM_LOG2E;
    const float ds = fdiv(-0.5f, dstd);
    __syncthreads();

    float spk[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float df = (float)k;
        spk[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fexp(fdiv(df * df, -FM_SQRT2 * sstd)))));
    }

    const int ox = tid & 15, oy = tid >> 4;                  // pixels (ox, oy) and (ox + 16, oy) of the sheared tile
    const int par = (K & 1) ? (oy & 1) : 0;                  // row parity: only matters for odd K
    // byte address of the centre of pixel A in plane 0 (dynamic LDS starts at the kernel's LDS base),
    // and the same shifted by one column either way for rows of odd parity
    uint32_t ctr = (uint32_t)(size_t)pl + (uint32_t)(((oy + HY) * RS + ox + M) * 4);
    uint32_t ctrp = ctr + 4u * (uint32_t)par, ctrm = ctr - 4u * (uint32_t)par;

    // One ds_read2_b32 brings the pair (plane[p], plane[p + 16]) into an aligned register pair.
    // The compiler's own pairing of LDS reads follows program order, not this pixel pairing, so
    // the reads are written out; they complete asynchronously and are waited for by the
    // "s_waitcnt lgkmcnt(0)" at the top of the step that consumes them (see STEP below).
#define RD2(dst, addr, o0) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(o0), "n"((o0) + 16) : "memory")
#define DELTA(r) (de_coff<PATTERN>(r, 1) - de_coff<PATTERN>(r, 0))
#define BASE(r) (DELTA(r) == 0 ? ctr : DELTA(r) > 0 ? ctrp : ctrm)
#define TAPOFF(r) (tap_dy<PATTERN>(r) * RS + de_coff<PATTERN>(r, 0))
    // pairs needed before the loop, read the ordinary way
    const float *c0 = pl + (oy + HY) * RS + ox + M;
#define CPAIR(k, r) ((f2){c0[TAPOFF(r) + par * DELTA(r) + (k) * LW], c0[TAPOFF(r) + par * DELTA(r) + (k) * LW + 16]})
    const f2 cw = CPAIR(3, 0);
    const f2 cfix = cw * (f2){frcp(cw.x + 1.0e-6f), frcp(cw.y + 1.0e-6f)};
    const f2 cx = CPAIR(0, 0) * cfix, cy = CPAIR(1, 0) * cfix, cz = CPAIR(2, 0) * cfix;
    const f2 cpow = CPAIR(4, 0);
    const bool liveA = cw.x > 0.0f, liveB = cw.y > 0.0f;
    f2 outx = 0.0f, outy = 0.0f, outz = 0.0f, outw = 0.0f, wsum = 0.0f;
    f2 wprev = CPAIR(3, -16);
    f2 px = CPAIR(0, -15), py = CPAIR(1, -15), pz = CPAIR(2, -15), pw = CPAIR(3, -15);
#undef CPAIR

    // Software pipeline, two taps per step, double-buffered: step g waits for the reads issued
    // in step g-1 (buffer g&1), issues the reads of step g+1, then does the arithmetic of its two
    // taps.  Every value that must not be touched early goes through the waiting asm.
    f2 L[2][2][6];                     // [buffer][tap][next x, next y, next z, next w, p, q]
    auto issue = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
#define ISSUE_TAP(k) if constexpr (-15 + g * 2 + (k) <= 15) { \
            constexpr int r = -15 + g * 2 + (k); \
            const uint32_t an = BASE(r + 1) + (uint32_t)(TAPOFF(r + 1) * 4); \
            const uint32_t ap = BASE(r) + (uint32_t)((TAPOFF(r) + 4 * LW) * 4); \
            RD2(L[g & 1][k][0], an, 0); RD2(L[g & 1][k][1], an, LW); RD2(L[g & 1][k][2], an, 2 * LW); RD2(L[g & 1][k][3], an, 3 * LW); \
            RD2(L[g & 1][k][4], ap, 0); RD2(L[g & 1][k][5], ap, LW); }
        ISSUE_TAP(0) ISSUE_TAP(1)
#undef ISSUE_TAP
    };
    auto step = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        f2 (&T)[2][6] = L[g & 1];
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(T[0][0]), "+v"(T[0][1]), "+v"(T[0][2]), "+v"(T[0][3]), "+v"(T[0][4]), "+v"(T[0][5]),
                       "+v"(T[1][0]), "+v"(T[1][1]), "+v"(T[1][2]), "+v"(T[1][3]), "+v"(T[1][4]), "+v"(T[1][5]),
                       "+v"(ctr), "+v"(ctrp), "+v"(ctrm), "+v"(wsum), "+v"(outx), "+v"(outy), "+v"(outz), "+v"(outw));
        if constexpr (g + 1 < 16) issue(std::integral_constant<int, g + 1>{});
        asm volatile("" : "+v"(px), "+v"(pw));      // the arithmetic below starts after the reads above are issued
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = -15 + g * 2 + k;
            if (r <= 15) {
                const f2 yd = px - cx, ud = py - cy, vd = pz - cz;
                f2 cdiff = f2fma(vd, vd, f2fma(ud, ud, yd * yd));
                cdiff.x = (pw.x > 0.0f && liveA) ? cdiff.x : 0.5f;
                cdiff.y = (pw.y > 0.0f && liveB) ? cdiff.y : 0.5f;
                const f2 ad = cpow - T[k][4];
                f2 e = f2fma((f2)ds, __builtin_elementwise_max(ad, -ad), cdiff * cs2);
                if (r != 0) {
                    const f2 gr = (T[k][3] - wprev) * T[k][5] * (r < 0 ? -gspeed : gspeed);
                    e -= (f2){fexp2(gr.x), fexp2(gr.y)};
                }
                const f2 factor = (f2){fexp2(e.x), fexp2(e.y)} * spk[r < 0 ? -r : r];
                wsum += factor;
                const f2 fw = factor * pw;
                outx = f2fma(fw, px, outx); outy = f2fma(fw, py, outy); outz = f2fma(fw, pz, outz); outw += fw;
                wprev = pw;
                px = T[k][0]; py = T[k][1]; pz = T[k][2]; pw = T[k][3];
            }
        }
    };
    issue(std::integral_constant<int, 0>{});
    [&]<int... G>(std::integer_sequence<int, G...>) __attribute__((always_inline)) {
        (step(std::integral_constant<int, G>{}), ...);
    }(std::make_integer_sequence<int, 16>{});
#undef RD2
#undef DELTA
#undef BASE
#undef TAPOFF
    const int xA = bx0 + ((oy * K) >> 1) + ox;               // image column of pixel A
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float ow = h ? outw.y : outw.x, ws = h ? wsum.y : wsum.x;
        const float sx = h ? outx.y : outx.x, sy = h ? outy.y : outy.x, sz = h ? outz.y : outz.x;
        const float wn = ow * frcp(ws + 1e-10f);
        const float rn = ow >= 1.17549435e-38f ? frcp(ow) : 0.0f;       // v_rcp_f32 of a denormal is +inf
        const int xo = xA + 16 * h;
        if (xo < 0 || xo >= (int)d.astride) continue;        // the parallelogram sticks out of the image at both ends of a band
        const uint32_t go = (uint32_t)((by0 + oy) * (int)d.astride + xo);
        Nout[go] = make_float4(sx * rn, sy * rn, sz * rn, wn);
        PRout[go].x = de_pow(wn, dpow);
        Wout[go] = wn;
    }
}

__global__ void __launch_bounds__(256)
k_de_prep2(fl_dim d, float4 *__restrict__ N, float2 *__restrict__ PR, float *__restrict__ W,
           const float4 *__restrict__ src, float dpow)
{
    PIX_IDX(d);
    const float4 p = src[gi];
    const float rw = p.w > 0.0f ? frcp(p.w) : 0.0f;
    N[gi] = make_float4(p.x * rw, p.y * rw, p.z * rw, p.w);
    PR[gi].x = de_pow(p.w, dpow);
    W[gi] = p.w;
}

// Both density blurs of one direction in one pass (cuburn/code/filters.py:106-131 as driven by
// cuburn/filters.py:80-84: 7 taps at step 1, then 7 taps at step 2 on the result), writing
// 1/(avg + 1e-6) into PR.y.  A workgroup stages the density tile plus both halos in LDS
// (edge-clamped), evaluates the first blur for the tile + second-blur halo, then the second.
// A first-blur value at a position outside the image is, by the clamped addressing of the
// two-kernel form, the first blur AT the clamped position; the summation order of each blur is
// that of k_den_blur / k_den_blur_1c, so the result is bit-identical to running them in turn.
#define DB_TW 64
#define DB_TH 16
template <int PATTERN, int STEP> __host__ __device__ __forceinline__ constexpr int db_hx() {
    int m = 0;
    for (int i = -3; i <= 3; ++i) { int v = tap_dx<PATTERN>(i * STEP); v = v < 0 ? -v : v; m = v > m ? v : m; }
    return m;
}
template <int PATTERN, int STEP> __host__ __device__ __forceinline__ constexpr int db_hy() {
    int m = 0;
    for (int i = -3; i <= 3; ++i) { int v = tap_dy<PATTERN>(i * STEP); v = v < 0 ? -v : v; m = v > m ? v : m; }
    return m;
}

template <int PATTERN>
__global__ void __launch_bounds__(256)
k_den_blur2_lds(fl_dim d, float2 *__restrict__ PR, const float *__restrict__ W, Coefs7 k)
{
    constexpr int H1X = db_hx<PATTERN, 1>(), H1Y = db_hy<PATTERN, 1>();
    constexpr int H2X = db_hx<PATTERN, 2>(), H2Y = db_hy<PATTERN, 2>();
    constexpr int SW = DB_TW + 2 * H2X, SH = DB_TH + 2 * H2Y;          // first-blur region
    constexpr int WW = SW + 2 * H1X, WH = SH + 2 * H1Y;                // density region
    __shared__ float sW[WH * WW];
    __shared__ float s1[SH * SW];
    const int tid = threadIdx.x;
    const int bx0 = blockIdx.x * DB_TW, by0 = blockIdx.y * DB_TH;
    const int xmax = (int)d.astride - 1, ymax = (int)d.ah - 1;

    // all global loads are issued before the first LDS store (a rolled loop pays one L2 round
    // trip per iteration)
    constexpr int NIT = (WH * WW + 255) / 256;
    float tw[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = min(it * 256 + tid, WH * WW - 1);
        const int ly = idx / WW, lx = idx - ly * WW;
        const int gx = min(max(bx0 + lx - H2X - H1X, 0), xmax), gy = min(max(by0 + ly - H2Y - H1Y, 0), ymax);
        tw[it] = W[(uint32_t)(gy * (int)d.astride + gx)];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * 256 + tid;
        if (idx < WH * WW) sW[idx] = tw[it];
    }
    __syncthreads();
    for (int idx = tid; idx < SH * SW; idx += 256) {
        const int qy = idx / SW, qx = idx - qy * SW;
        // local density-region coordinates of the CLAMPED position
        const int lx = min(max(bx0 + qx - H2X, 0), xmax) - bx0 + H2X + H1X;
        const int ly = min(max(by0 + qy - H2Y, 0), ymax) - by0 + H2Y + H1Y;
        float den = 0.0f;
#pragma unroll
        for (int i = 0; i < 7; ++i)
            den += sW[(ly + tap_dy<PATTERN>(i - 3)) * WW + lx + tap_dx<PATTERN>(i - 3)] * k.c[i];
        s1[idx] = den;
    }
    __syncthreads();
    for (int o = tid; o < DB_TW * DB_TH; o += 256) {
        const int oy = o / DB_TW, ox = o - oy * DB_TW;
        if (bx0 + ox > xmax) continue;
        float den = 0.0f;
#pragma unroll
        for (int i = 0; i < 7; ++i)
            den += s1[(oy + H2Y + tap_dy<PATTERN>(2 * (i - 3))) * SW + ox + H2X + tap_dx<PATTERN>(2 * (i - 3))] * k.c[i];
        PR[(uint32_t)((by0 + oy) * (int)d.astride + bx0 + ox)].y = frcp(den + 1.0e-6f);
    }
}

__global__ void __launch_bounds__(256) k_logscale(fl_dim d, float4 *__restrict__ buf, float k1, float k2) {
    PIX_IDX(d);
    buf[gi] = logscale_px(buf[gi], k1, k2);
}

__global__ void __launch_bounds__(256)
k_colorclip(fl_dim d, float4 *__restrict__ buf, float vib, float highpow, float gam, float lin, float lingam) {
    PIX_IDX(d);
    buf[gi] = colorclip_px(buf[gi], vib, highpow, gam, lin, lingam);
}

// Fused ends of the default chains (the ABI defers `yuv` and the un-normalising step of the DE
// so that the next filter call can take them along; every pixel goes through the same device
// functions in the same order as in the separate kernels: bit-identical results).
//   yuv -> DE prep:  accumulator -> N, PR.x, W in one pass (saves a float4 round trip)
__global__ void __launch_bounds__(256)
k_yuv_de_prep2(fl_dim d, float4 *__restrict__ N, float2 *__restrict__ PR, float *__restrict__ W,
               const float4 *__restrict__ src, float dpow)
{
    PIX_IDX(d);
    const float4 p = yuv_px(src[gi]);
    const float rw = p.w > 0.0f ? frcp(p.w) : 0.0f;
    N[gi] = make_float4(p.x * rw, p.y * rw, p.z * rw, p.w);
    PR[gi].x = de_pow(p.w, dpow);
    W[gi] = p.w;
}
//   DE finish [-> logscale] [-> colorclip] in one pass
__global__ void __launch_bounds__(256)
k_de_finish_tone(fl_dim d, float4 *__restrict__ dst, const float4 *__restrict__ N, int do_log, float k1, float k2,
                 int do_clip, float vib, float highpow, float gam, float lin, float lingam)
{
    PIX_IDX(d);
    const float4 n = N[gi];
    float4 p = make_float4(n.x * n.w, n.y * n.w, n.z * n.w, n.w);
    if (do_log) p = logscale_px(p, k1, k2);
    if (do_clip) p = colorclip_px(p, vib, highpow, gam, lin, lingam);
    dst[gi] = p;
}

// cuburn/code/filters.py:294-302
__global__ void __launch_bounds__(256) k_gamma_full_hi(fl_dim d, float4 *__restrict__ dst, const float4 *__restrict__ src) {
    PIX_IDX(d);
    float4 p = src[gi];
    float ls = 0.0f;
    if (p.w > 0.0f) ls = fdiv(fmaxf(0.0f, p.w - 1.0f), p.w);
    p.x *= ls; p.y *= ls; p.z *= ls; p.w *= ls;
    dst[gi] = p;
}

__device__ __forceinline__ float clip_ls(float w, float gam_m_1, float lin, float lingam) {
    float ls = fpow(w, gam_m_1);
    if (w < lin) {
        const float frac = fdiv(w, lin);
        ls = (1.0f - frac) * lingam + frac * ls;
    }
    return ls;
}

// cuburn/code/filters.py:304-328
__global__ void __launch_bounds__(256)
k_smearclip(fl_dim d, float4 *__restrict__ buf, const float4 *__restrict__ smear, float gam_m_1, float lin, float lingam) {
    PIX_IDX(d);
    float4 p = buf[gi];
    const float4 a = smear[gi];
    p.x += a.x; p.y += a.y; p.z += a.z; p.w += a.w;
    if (p.w <= 0.0f) { buf[gi] = make_float4(0, 0, 0, 0); return; }
    const float ls = clip_ls(p.w, gam_m_1, lin, lingam);
    p.x *= ls; p.y *= ls; p.z *= ls; p.w *= ls;
    buf[gi] = p;
}

// cuburn/code/filters.py:268-272 (reads pix.x)
__global__ void __launch_bounds__(256) k_apply_gamma(fl_dim d, float *__restrict__ dst, const float4 *__restrict__ src, float gamma) {
    PIX_IDX(d);
    dst[gi] = fpow(src[gi].x, gamma);
}

// cuburn/code/filters.py:274-288
__global__ void __launch_bounds__(256) k_haloclip(fl_dim d, float4 *__restrict__ buf, const float *__restrict__ den, float gam_m_1) {
    PIX_IDX(d);
    float4 p = buf[gi];
    if (p.w <= 0.0f) { buf[gi] = make_float4(0, 0, 0, 0); return; }
    const float ls = fdiv(fpow(p.w, gam_m_1), fmaxf(1.0f, den[gi]));
    p.x *= ls; p.y *= ls; p.z *= ls; p.w *= ls;
    buf[gi] = p;
}

// cuburn/code/filters.py:332-350
__global__ void __launch_bounds__(256)
k_plainclip(fl_dim d, float4 *__restrict__ buf, float gam_m_1, float lin, float lingam, float brightness) {
    PIX_IDX(d);
    float4 p = buf[gi];
    if (p.w <= 0.0f) { buf[gi] = make_float4(0, 0, 0, 0); return; }
    const float ls = clip_ls(p.w, gam_m_1, lin, lingam) * brightness;
    p.x *= ls; p.y *= ls; p.z *= ls; p.w *= ls;
    buf[gi] = p;
}

// cuburn/code/filters.py:81-90
__global__ void __launch_bounds__(256) k_logencode(fl_dim d, float4 *__restrict__ dst, const float4 *__restrict__ src, float degamma) {
    PIX_IDX(d);
    float4 p = src[gi];
    p.x = flog2(fpow(p.x, degamma)) * (1.0f / 12.0f) + 1.0f;
    p.y = flog2(fpow(p.y, degamma)) * (1.0f / 12.0f) + 1.0f;
    p.z = flog2(fpow(p.z, degamma)) * (1.0f / 12.0f) + 1.0f;
    p.w = flog2(fpow(p.w, degamma)) * (1.0f / 12.0f) + 1.0f;
    dst[gi] = p;
}

// ---- launchers: cuburn/code/util.py:45-53 launch2 grid (astride/32, ah/8) x (32,8) ----------
#define GRID(d) dim3((d).astride / 32, (d).ah / 8), dim3(32, 8)
static Coefs7 mk(const float *c) { Coefs7 k; for (int i = 0; i < 7; ++i) k.c[i] = c[i]; return k; }

void launch_yuv_to_rgb(hipStream_t st, fl_dim d, float4 *dst, const float4 *src) { hipLaunchKernelGGL(k_yuv_to_rgb, GRID(d), 0, st, d, dst, src); }
void launch_den_blur(hipStream_t st, fl_dim d, float *dst, const float4 *src, int p, int up, const float *c) { hipLaunchKernelGGL(k_den_blur, GRID(d), 0, st, d, dst, src, p, up, mk(c)); }
void launch_den_blur_1c(hipStream_t st, fl_dim d, float *dst, const float *src, int p, int up, const float *c) { hipLaunchKernelGGL(k_den_blur_1c, GRID(d), 0, st, d, dst, src, p, up, mk(c)); }
void launch_full_blur(hipStream_t st, fl_dim d, float4 *dst, const float4 *src, int p, int up, const float *c) { hipLaunchKernelGGL(k_full_blur, GRID(d), 0, st, d, dst, src, p, up, mk(c)); }
void launch_bilateral(hipStream_t st, fl_dim d, float4 *dst, const float4 *src, const float *blur, int pattern, int radius,
                      float sstd, float cstd, float dstd, float dpow, float gspeed) {
    hipLaunchKernelGGL(k_bilateral, GRID(d), 0, st, d, dst, src, blur, pattern, radius, sstd, cstd, dstd, dpow, gspeed);
}
void launch_de_prep2(hipStream_t st, fl_dim d, float4 *N, float *PR, float *W, const float4 *src, float dpow) { hipLaunchKernelGGL(k_de_prep2, GRID(d), 0, st, d, N, (float2 *)PR, W, src, dpow); }
void launch_den_blur2_lds(hipStream_t st, fl_dim d, int pattern, float *PR, const float *W, const float *c) {
#define DB(P) case P: hipLaunchKernelGGL(k_den_blur2_lds<P>, dim3((d.astride + DB_TW - 1) / DB_TW, d.ah / DB_TH), dim3(256), 0, st, d, (float2 *)PR, W, mk(c)); break
    switch (pattern) { DB(0); DB(1); DB(2); DB(3); DB(4); DB(5); DB(6); DB(7); default: break; }
#undef DB
}
template <int P>
static void launch_de_pk_one(hipStream_t st, fl_dim d, float4 *Nout, float2 *PRout, float *Wout, const float4 *N, const float2 *PR,
                             float sstd, float cstd, float dstd, float dpow, float gspeed) {
    constexpr int LH = DE_TH + 2 * de_hy<P>();
    const size_t lds = (size_t)LH * de_pk_row_stride<P>() * 4;
    static unsigned long long attr = 0;
    ensure_max_dynamic_lds((const void *)k_de_bilateral_pk<P>, attr);
    hipLaunchKernelGGL(k_de_bilateral_pk<P>, dim3((d.astride + de_span<P>() + DE_TW - 1) / DE_TW, d.ah / DE_TH), dim3(256), lds, st,
                       d, Nout, PRout, Wout, N, PR, sstd, cstd, dstd, dpow, gspeed);
}
void launch_de_bilateral_lds(hipStream_t st, fl_dim d, int pattern, float4 *Nout, float *PRout, float *Wout, const float4 *N, const float *PR,
                             float sstd, float cstd, float dstd, float dpow, float gspeed) {
#define DE(P) case P: launch_de_pk_one<P>(st, d, Nout, (float2 *)PRout, Wout, N, (const float2 *)PR, sstd, cstd, dstd, dpow, gspeed); break
    switch (pattern) { DE(0); DE(1); DE(2); DE(3); DE(4); DE(5); DE(6); DE(7); default: break; }
#undef DE
}
void launch_yuv_de_prep2(hipStream_t st, fl_dim d, float4 *N, float *PR, float *W, const float4 *src, float dpow) { hipLaunchKernelGGL(k_yuv_de_prep2, GRID(d), 0, st, d, N, (float2 *)PR, W, src, dpow); }
void launch_de_finish_tone(hipStream_t st, fl_dim d, float4 *dst, const float4 *N, bool do_log, float k1, float k2, bool do_clip, const float *cc) {
    hipLaunchKernelGGL(k_de_finish_tone, GRID(d), 0, st, d, dst, N, do_log ? 1 : 0, k1, k2, do_clip ? 1 : 0, do_clip ? cc[0] : 0.f, do_clip ? cc[1] : 0.f, do_clip ? cc[2] : 0.f, do_clip ? cc[3] : 0.f, do_clip ? cc[4] : 0.f);
}
void launch_logscale(hipStream_t st, fl_dim d, float4 *buf, float k1, float k2) { hipLaunchKernelGGL(k_logscale, GRID(d), 0, st, d, buf, k1, k2); }
void launch_colorclip(hipStream_t st, fl_dim d, float4 *buf, float vib, float hp, float gam, float lin, float lingam) { hipLaunchKernelGGL(k_colorclip, GRID(d), 0, st, d, buf, vib, hp, gam, lin, lingam); }
void launch_gamma_full_hi(hipStream_t st, fl_dim d, float4 *dst, const float4 *src) { hipLaunchKernelGGL(k_gamma_full_hi, GRID(d), 0, st, d, dst, src); }
void launch_smearclip(hipStream_t st, fl_dim d, float4 *buf, const float4 *smear, float g, float lin, float lingam) { hipLaunchKernelGGL(k_smearclip, GRID(d), 0, st, d, buf, smear, g, lin, lingam); }
void launch_apply_gamma(hipStream_t st, fl_dim d, float *dst, const float4 *src, float gamma) { hipLaunchKernelGGL(k_apply_gamma, GRID(d), 0, st, d, dst, src, gamma); }
void launch_haloclip(hipStream_t st, fl_dim d, float4 *buf, const float *den, float g) { hipLaunchKernelGGL(k_haloclip, GRID(d), 0, st, d, buf, den, g); }
void launch_plainclip(hipStream_t st, fl_dim d, float4 *buf, float g, float lin, float lingam, float b) { hipLaunchKernelGGL(k_plainclip, GRID(d), 0, st, d, buf, g, lin, lingam, b); }
void launch_logencode(hipStream_t st, fl_dim d, float4 *dst, const float4 *src, float degamma) { hipLaunchKernelGGL(k_logencode, GRID(d), 0, st, d, dst, src, degamma); }
