// output.hip — float4 -> rgba8 / rgba16 / planar YUV with dithering and gutter crop.
// Device side of cuburn/code/output.py:7-236 (dclampf, f32_to_rgba_u8, f32_to_rgba_u16,
// f32_to_yuv444p, f32_to_yuv444p10, f32_to_yuv420p10, f32_to_yuv444p12).
// RNG use: state t serves output pixels t, t+nrng, t+2*nrng, ... so that no state is ever
// shared by two live threads (the reference's ring buffer can hand one slot to two blocks).
#include "flame_device.h"
#include "kernels.h"

__device__ __forceinline__ float dclampf(mwc_t &r, float peak, float in) {
    float ret = 0.0f;
    if (in > 0.0f) ret = fminf(peak, in * peak + 0.99f * mwc_next_01(r));
    return ret;
}

template <typename T4, typename T, int PEAK>
__global__ void __launch_bounds__(256)
k_f32_to_rgba(fl_dim d, const float4 *__restrict__ src, fl_mwc *__restrict__ rng, uint32_t nrng, T4 *__restrict__ dst)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nrng) return;
    mwc_t r = {rng[t].mul, rng[t].state, rng[t].carry};
    const uint32_t npix = d.w * d.h;
    // Only nrng threads exist (one RNG state each, 4 waves per CU), so the kernel is bound by the
    // latency of its loads: four pixels are requested at a time (eight measure the same), then dithered in order (the draws of a
    // state are consumed in the same order as before).
    for (uint32_t p0 = t; p0 < npix; p0 += 4u * nrng) {
        float4 in[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t p = p0 + k * nrng, pc = p < npix ? p : t;
            const uint32_t x = pc % d.w, y = pc / d.w;
            in[k] = src[(size_t)d.astride * (y + FL_GUTTER) + x + FL_GUTTER];        // output.py:28
        }
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t p = p0 + k * nrng;
            if (p < npix) {
                T4 o;
                o.x = (T)dclampf(r, (float)PEAK, in[k].x);
                o.y = (T)dclampf(r, (float)PEAK, in[k].y);
                o.z = (T)dclampf(r, (float)PEAK, in[k].z);
                o.w = (T)dclampf(r, (float)PEAK, in[k].w);
                dst[p] = o;
            }
        }
    }
    rng[t].mul = r.mul; rng[t].state = r.state; rng[t].carry = r.carry;
}

__device__ __forceinline__ unsigned short sat_u16(float v) { return v >= 65535.0f ? 65535 : v > 0.0f ? (unsigned short)v : 0; }
__device__ __forceinline__ float yuv_cb(float4 in) { return -0.168736f * in.x - 0.331264f * in.y + 0.5f * in.z; }
__device__ __forceinline__ float yuv_cr(float4 in) { return 0.5f * in.x - 0.418688f * in.y - 0.081312f * in.z; }

// Planar YUV for the video encoders (cuburn/code/output.py:75-221; formats as in flame_hip.h
// FL_OUT_*).  A pixel draws for Y, Cb, Cr in that order; in 4:2:0 the thread of pixel (x, y) of
// the top-left quadrant also produces chroma sample (x, y) — the alpha-weighted mean of its 2x2
// source pixels — as the reference's kernel does (:157-183).  Reference quirks kept: the 10-bit
// 4:4:4 Cb plane is stored undithered although its draw is made (:122,129).
template <int FMT>
__global__ void __launch_bounds__(256)
k_f32_to_yuv(fl_dim d, const float4 *__restrict__ src, fl_mwc *__restrict__ rng, uint32_t nrng, void *__restrict__ dstv)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nrng) return;
    mwc_t r = {rng[t].mul, rng[t].state, rng[t].carry};
    const uint32_t npix = d.w * d.h;
    unsigned char *d8 = (unsigned char *)dstv;
    unsigned short *d16 = (unsigned short *)dstv;
    for (uint32_t p = t; p < npix; p += nrng) {
        const uint32_t x = p % d.w, y = p / d.w;
        const float4 in = src[(size_t)d.astride * (y + FL_GUTTER) + x + FL_GUTTER];
        if (FMT == FL_OUT_YUV444P || FMT == FL_OUT_YUV444P10) {
            const float peak = FMT == FL_OUT_YUV444P ? 255.0f : 1023.0f;
            const float cb = yuv_cb(in) + 0.5f;
            const float fy = dclampf(r, peak, 0.299f * in.x + 0.587f * in.y + 0.114f * in.z);
            const float fb = dclampf(r, peak, cb);
            const float fr = dclampf(r, peak, yuv_cr(in) + 0.5f);
            if (FMT == FL_OUT_YUV444P) { d8[p] = (unsigned char)fy; d8[npix + p] = (unsigned char)fb; d8[2 * npix + p] = (unsigned char)fr; }
            else { d16[p] = (unsigned short)fy; d16[npix + p] = sat_u16(1023.0f * cb); d16[2 * npix + p] = (unsigned short)fr; }
        } else if (FMT == FL_OUT_YUV420P10) {
            d16[p] = (unsigned short)dclampf(r, 1023.0f, 0.299f * in.x + 0.587f * in.y + 0.114f * in.z);
            if (x < d.w / 2 && y < d.h / 2) {
                const float4 *q = src + (size_t)d.astride * (2 * y + FL_GUTTER) + 2 * x + FL_GUTTER;
                const float4 q0 = q[0], q1 = q[1], q2 = q[d.astride], q3 = q[d.astride + 1];
                float sum = (float)((double)q0.w + 1e-12), cb = q0.w * yuv_cb(q0), cr = q0.w * yuv_cr(q0);
                sum += q1.w; cb += q1.w * yuv_cb(q1); cr += q1.w * yuv_cr(q1);
                sum += q2.w; cb += q2.w * yuv_cb(q2); cr += q2.w * yuv_cr(q2);
                sum += q3.w; cb += q3.w * yuv_cb(q3); cr += q3.w * yuv_cr(q3);
                const uint32_t c = (d.w / 2) * y + x;
                d16[npix + c] = (unsigned short)dclampf(r, 1023.0f, cb / sum + 0.5f);
                d16[npix + npix / 4 + c] = (unsigned short)dclampf(r, 1023.0f, cr / sum + 0.5f);
            }
        } else {
            const float cx = fminf(1.0f, fmaxf(0.0f, in.x)), cy = fminf(1.0f, fmaxf(0.0f, in.y)), cz = fminf(1.0f, fmaxf(0.0f, in.z));
            d16[p] = (unsigned short)(dclampf(r, 3504.0f, 0.2126f * cx + 0.7152f * cy + 0.0722f * cz) + 256.0f);
            d16[npix + p] = (unsigned short)(dclampf(r, 3584.0f, -0.11457f * cx - 0.38543f * cy + 0.5f * cz + 0.5f) + 256.0f);
            d16[2 * npix + p] = (unsigned short)(dclampf(r, 3584.0f, 0.5f * cx - 0.45416f * cy - 0.04585f * cz + 0.5f) + 256.0f);
        }
    }
    rng[t].mul = r.mul; rng[t].state = r.state; rng[t].carry = r.carry;
}

void launch_f32_to_rgba(hipStream_t st, fl_dim d, const float4 *src, fl_mwc *rng, uint32_t nrng, int fmt, void *dst)
{
    dim3 grid((nrng + 255) / 256), block(256);
    switch (fmt) {
    case FL_OUT_YUV444P: hipLaunchKernelGGL((k_f32_to_yuv<FL_OUT_YUV444P>), grid, block, 0, st, d, src, rng, nrng, dst); return;
    case FL_OUT_YUV444P10: hipLaunchKernelGGL((k_f32_to_yuv<FL_OUT_YUV444P10>), grid, block, 0, st, d, src, rng, nrng, dst); return;
    case FL_OUT_YUV420P10: hipLaunchKernelGGL((k_f32_to_yuv<FL_OUT_YUV420P10>), grid, block, 0, st, d, src, rng, nrng, dst); return;
    case FL_OUT_YUV444P12: hipLaunchKernelGGL((k_f32_to_yuv<FL_OUT_YUV444P12>), grid, block, 0, st, d, src, rng, nrng, dst); return;
    default: break;
    }
    if (fmt == 0) hipLaunchKernelGGL((k_f32_to_rgba<uchar4, unsigned char, 255>), grid, block, 0, st, d, src, rng, nrng, (uchar4 *)dst);
    else hipLaunchKernelGGL((k_f32_to_rgba<ushort4, unsigned short, 65535>), grid, block, 0, st, d, src, rng, nrng, (ushort4 *)dst);
}

// ---- measurement: the streaming copy rate of the device (fl_measure_copy) -------------------------------------------
// One float4 per thread, non-temporal both ways (the data is touched once), consecutive workgroups on consecutive 4 KB: 6.2-6.4 TB/s.
// (tools/copy_bench.hip, profiles/r06_copy_bench.txt: the same copy as a grid-stride loop of 1024-65536 workgroups runs at 4.4-5.4 TB/s —
// every workgroup then strides through the whole GiB —, hipMemcpyDtoD and torch's copy_ at 5.2-5.5.)
typedef float f32x4_ __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy_nt(f32x4_ *__restrict__ dst, const f32x4_ *__restrict__ src, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

int launch_measure_copy(size_t nbytes, int iters, float *ms)
{
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = -1;
    const size_t n4 = nbytes / 16;
    const unsigned blocks = (unsigned)((n4 + 255) / 256);
    if (n4 / 256 < 0x7fffffffu && hipMalloc(&a, n4 * 16) == hipSuccess && hipMalloc(&b, n4 * 16) == hipSuccess && hipMemset(a, 1, n4 * 16) == hipSuccess &&
        hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(k_copy_nt, dim3(blocks), dim3(256), 0, 0, (f32x4_ *)b, (const f32x4_ *)a, n4);
        hipEventRecord(e0, 0);
        for (int k = 0; k < iters; ++k) hipLaunchKernelGGL(k_copy_nt, dim3(blocks), dim3(256), 0, 0, (f32x4_ *)b, (const f32x4_ *)a, n4);
        hipEventRecord(e1, 0);
        float t = 0.0f;
        if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&t, e0, e1) == hipSuccess) { *ms = t / (float)iters; rc = 0; }
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(a); hipFree(b);
    (void)hipGetLastError();
    return rc;
}
