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

__global__ void __launch_bounds__(256) k_logscale(fl_dim d, float4 *__restrict__ buf, float k1, float k2) {
    PIX_IDX(d);
    buf[gi] = logscale_px(buf[gi], k1, k2);
}

__global__ void __launch_bounds__(256)
k_colorclip(fl_dim d, float4 *__restrict__ buf, float vib, float highpow, float gam, float lin, float lingam) {
    PIX_IDX(d);
    buf[gi] = colorclip_px(buf[gi], vib, highpow, gam, lin, lingam);
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
void launch_den_blur_1c(hipStream_t st, fl_dim d, float *dst, const float *src, int p, int up, const float *c) { hipLaunchKernelGGL(k_den_blur_1c, GRID(d), 0, st, d, dst, src, p, up, mk(c)); }
void launch_full_blur(hipStream_t st, fl_dim d, float4 *dst, const float4 *src, int p, int up, const float *c) { hipLaunchKernelGGL(k_full_blur, GRID(d), 0, st, d, dst, src, p, up, mk(c)); }
void launch_logscale(hipStream_t st, fl_dim d, float4 *buf, float k1, float k2) { hipLaunchKernelGGL(k_logscale, GRID(d), 0, st, d, buf, k1, k2); }
void launch_colorclip(hipStream_t st, fl_dim d, float4 *buf, float vib, float hp, float gam, float lin, float lingam) { hipLaunchKernelGGL(k_colorclip, GRID(d), 0, st, d, buf, vib, hp, gam, lin, lingam); }
void launch_gamma_full_hi(hipStream_t st, fl_dim d, float4 *dst, const float4 *src) { hipLaunchKernelGGL(k_gamma_full_hi, GRID(d), 0, st, d, dst, src); }
void launch_smearclip(hipStream_t st, fl_dim d, float4 *buf, const float4 *smear, float g, float lin, float lingam) { hipLaunchKernelGGL(k_smearclip, GRID(d), 0, st, d, buf, smear, g, lin, lingam); }
void launch_apply_gamma(hipStream_t st, fl_dim d, float *dst, const float4 *src, float gamma) { hipLaunchKernelGGL(k_apply_gamma, GRID(d), 0, st, d, dst, src, gamma); }
void launch_haloclip(hipStream_t st, fl_dim d, float4 *buf, const float *den, float g) { hipLaunchKernelGGL(k_haloclip, GRID(d), 0, st, d, buf, den, g); }
void launch_plainclip(hipStream_t st, fl_dim d, float4 *buf, float g, float lin, float lingam, float b) { hipLaunchKernelGGL(k_plainclip, GRID(d), 0, st, d, buf, g, lin, lingam, b); }
void launch_logencode(hipStream_t st, fl_dim d, float4 *dst, const float4 *src, float degamma) { hipLaunchKernelGGL(k_logencode, GRID(d), 0, st, d, dst, src, degamma); }
