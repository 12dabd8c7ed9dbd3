// interp.hip — per-frame parameter preparation on device (tiny kernels).
//
//  * k_interp_params : cuburn/code/interp.py:234-272 (interp_iter_params) + the precalc
//    snippets of cuburn/code/iter.py:12-30,56-95 and cuburn/code/variations.py
//    (:136-140, :267-273, :292-294, :630-634), driven by the op list of
//    include/flame_hip.h (6) instead of generated code.  One thread per (temporal sample, op);
//    there is one temporal sample per walker slot (nts = nslots, see iter.hip).
//  * k_interp_palette: cuburn/code/interp.py:372-433 (interp_color + interp_palette_flat);
//    writes the packed-u64 palette (256 x 64) into plain global memory (the reference's
//    CUDA surface has no CDNA equivalent; the iterate kernel stages its row in LDS).
#include "flame_device.h"
#include "kernels.h"

// cuburn/code/util.py:219-230: rightmost index whose value is strictly below the needle
__device__ __forceinline__ int binsearch32(const float *hay, float needle) {
    int lo = 0;
#pragma unroll
    for (int i = 4; i >= 0; --i)
        if (needle > hay[lo + (1 << i)]) lo += 1 << i;
    return lo;
}

#define ELBOW 0.0625f
#define ELOG1 5.0f
__device__ __forceinline__ float linlog(float x) {
    if (x > ELBOW) return log2f(x) + ELOG1;
    if (x < -ELBOW) return -(log2f(-x) + ELOG1);
    return x / ELBOW;
}
__device__ __forceinline__ float linexp(float v) {
    if (v >= 1.0f) return exp2f(v - ELOG1);
    if (v <= -1.0f) return -exp2f(-v - ELOG1);
    return v * ELBOW;
}
__device__ __forceinline__ float linslope(float x, float m) {
    if (x >= ELBOW) return m / x;
    if (x <= -ELBOW) return m / -x;
    return m / ELBOW;
}

// cuburn/code/interp.py:318-355
__device__ float catmull_rom(const float *times, const float *knots, float t, bool mag) {
    int idx = max(binsearch32(times, t), 1);
    float t1 = times[idx], t2 = times[idx + 1] - t1;
    float rt2 = 1.0f / t2;
    float t0 = (times[idx - 1] - t1) * rt2, t3 = (times[idx + 2] - t1) * rt2;
    t = (t - t1) * rt2;
    float k0 = knots[idx - 1], k1 = knots[idx], k2 = knots[idx + 1], k3 = knots[idx + 2];
    float m1 = (k2 - k0) / (1.0f - t0), m2 = (k3 - k1) / t3;
    if (mag) {
        m1 = linslope(k1, m1);
        m2 = linslope(k2, m2);
        k1 = linlog(k1);
        k2 = linlog(k2);
    }
    float tt = t * t, ttt = tt * t;
    float r = m1 * (ttt - 2.0f * tt + t) + k1 * (2.0f * ttt - 3.0f * tt + 1.0f)
            + m2 * (ttt - tt) + k2 * (-2.0f * ttt + 3.0f * tt);
    return mag ? linexp(r) : r;
}

__global__ void __launch_bounds__(256)
k_interp_params(float *__restrict__ params, const float *__restrict__ times, const float *__restrict__ knots,
                const int4 *__restrict__ ops, uint32_t nops, uint32_t pstride, uint32_t nts, float tstart, float tstep,
                fl_dim dim)
{
    // grid (temporal samples / 256, ops): one thread evaluates ONE op of one temporal sample (ops
    // write disjoint words of the block; the launcher zero-fills the padding first).  One thread
    // per sample walking the whole op list was a 40 us serial chain at the head of every frame.
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= nts) return;
    const float time = tstart + (float)id * tstep;
    float *out = params + (size_t)id * pstride;
#define ROW(r, mag) catmull_rom(times + (size_t)(r) * FL_KNOTS, knots + (size_t)(r) * FL_KNOTS, time, mag)
    {
        const int4 op = ops[blockIdx.y];
        float *o = out + op.y;
        switch (op.x) {
        case FL_OP_SPLINE: o[0] = ROW(op.z, false); break;
        case FL_OP_SPLINE_MAG: o[0] = ROW(op.z, true); break;
        case FL_OP_CAMERA: {            // cuburn/code/iter.py:56-79
            float rot = ROW(op.z, false) * FM_PI / 180.0f;
            float rs = sinf(rot), rc = cosf(rot);
            float cenx = ROW(op.z + 1, false), ceny = ROW(op.z + 2, false);
            float scale = ROW(op.z + 3, true) * (float)dim.w;
            o[0] = scale * rc;
            o[1] = scale * -rs;
            o[2] = scale * (rs * ceny - rc * cenx) + 0.5f * (float)dim.aw;
            o[3] = scale * rs;
            o[4] = scale * rc;
            o[5] = scale * -(rs * cenx + rc * ceny) + 0.5f * (float)dim.ah;
        } break;
        case FL_OP_AFFINE: {            // cuburn/code/iter.py:81-95
            float pri = ROW(op.z, false) * FM_PI / 180.0f;
            float spr = ROW(op.z + 1, false) * FM_PI / 180.0f;
            float magx = ROW(op.z + 2, true), magy = ROW(op.z + 3, true);
            o[0] = magx * cosf(pri - spr);       // xx
            o[3] = -magx * sinf(pri - spr);      // yx
            o[1] = -magy * cosf(pri + spr);      // xy
            o[4] = magy * sinf(pri + spr);       // yy
            o[2] = ROW(op.z + 4, false);         // xo
            o[5] = -ROW(op.z + 5, false);        // yo
        } break;
        case FL_OP_CDF: {               // cuburn/code/iter.py:12-30
            float sum = 0.0f;
            for (int k = 0; k < op.w; ++k) sum += ROW(op.z + k, false);
            float rsum = 1.0f / sum;
            sum = 0.0f;
            for (int k = 0; k < op.w; ++k) { sum += ROW(op.z + k, false) * rsum; o[k] = sum; }
            o[op.w - 1] = 2.0f;         // the last xform takes everything that is left
        } break;
        case FL_OP_RATIO2: o[0] = ROW(op.z, true) / (2.0f * ROW(op.w, true)); break;
        case FL_OP_INVSQ: { float v = ROW(op.z, false); o[0] = 1.0f / (v * v + 1.0e-20f); } break;
        case FL_OP_INVSQ_MAX: { float v = ROW(op.z, true); o[0] = 1.0f / fmaxf(1e-20f, v * v); } break;
        case FL_OP_PERSP: {             // cuburn/code/variations.py:267-273
            float pang = ROW(op.z, false) * FM_PI_2;
            float pdist = fmaxf(1e-9f, ROW(op.w, true));
            o[0] = pdist; o[1] = sinf(pang); o[2] = pdist * cosf(pang);
        } break;
        case FL_OP_CONST: o[0] = __int_as_float(op.z); break;
        default: break;
        }
    }
#undef ROW
}

__device__ __forceinline__ float3 rgb2yuv(float3 c) {          // cuburn/code/color.py:18-23
    return make_float3(0.299f * c.x + 0.587f * c.y + 0.114f * c.z,
                       -0.168736f * c.x - 0.331264f * c.y + 0.5f * c.z,
                       0.5f * c.x - 0.418688f * c.y - 0.081312f * c.z);
}
// `uint32_t y = f` of the reference: truncate toward zero, negatives / NaN -> 0
__device__ __forceinline__ uint32_t f2u_trunc(float f) { return (f > 0.0f) ? (uint32_t)fminf(f, 4294967040.0f) : 0u; }

__global__ void __launch_bounds__(FL_PAL_W)
k_interp_palette(fl_mwc *__restrict__ rngs, const float *__restrict__ ptimes, const float4 *__restrict__ pals,
                 float tstart, float tstep, u64 *__restrict__ out)
{
    const uint32_t c = threadIdx.x, row = blockIdx.x;
    fl_mwc *rp = rngs + row * FL_PAL_W + c;
    mwc_t r = {rp->mul, rp->state, rp->carry};
    const float time = tstart + (float)row * tstep;
    int idx = (int)fmaxf((float)(binsearch32(ptimes, time) + 1), 1.0f);
    float tr = ptimes[idx];
    float lf = (tr - time) / (tr - ptimes[idx - 1]);
    float rf = 1.0f - lf;
    float4 left = pals[FL_PAL_W * (idx - 1) + c];
    float4 right = tr > 1.0f ? left : pals[FL_PAL_W * idx + c];
    if (tr > 1.0f) { lf = 1.0f; rf = 0.0f; }
    float3 ly = rgb2yuv(make_float3(left.x, left.y, left.z));
    float3 ry = rgb2yuv(make_float3(right.x, right.y, right.z));
    float Y = ly.x * lf + ry.x * rf, U = ly.y * lf + ry.y * rf + 0.5f, V = ly.z * lf + ry.z * rf + 0.5f;
    uint32_t y = f2u_trunc(Y * 255.0f + 0.49f * mwc_next_11(r));
    uint32_t u = f2u_trunc(U * 255.0f + 0.49f * mwc_next_11(r));
    uint32_t v = f2u_trunc(V * 255.0f + 0.49f * mwc_next_11(r));
    y = min(255u, y); u = min(255u, u); v = min(255u, v);
    const uint32_t hi = (1u << 22) | (y << 4), lo = (u << 18) | v;
    out[row * FL_PAL_W + c] = ((u64)hi << 32) | lo;
    rp->mul = r.mul; rp->state = r.state; rp->carry = r.carry;
}

void launch_interp_palette(hipStream_t st, fl_mwc *rng_pal, const float *ptimes, const float4 *pals,
                           float ts, float tstep, u64 *out)
{
    hipLaunchKernelGGL(k_interp_palette, dim3(FL_PAL_H), dim3(FL_PAL_W), 0, st, rng_pal, ptimes, pals, ts, tstep, out);
}

void launch_interp_params(hipStream_t st, float *params, const float *times, const float *knots,
                          const int32_t *ops, uint32_t nops, uint32_t pstride, uint32_t nts, float ts, float tstep, fl_dim dim,
                          bool zero_first)
{
    // padding / unused post affines read as zero.  The ops of a genome write the same words every
    // frame, so the fill is only needed when the blocks last held another genome's parameters.
    if (zero_first) hipMemsetAsync(params, 0, sizeof(float) * (size_t)nts * pstride, st);
    if (nops == 0) return;
    hipLaunchKernelGGL(k_interp_params, dim3((nts + 255) / 256, nops), dim3(256), 0, st, params, times, knots,
                       (const int4 *)ops, nops, pstride, nts, ts, tstep, dim);
}
