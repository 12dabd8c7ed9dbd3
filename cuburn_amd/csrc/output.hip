// output.hip — float4 -> rgba8 / rgba16 with dithering and gutter crop.
// Device side of cuburn/code/output.py:7-71 (dclampf, f32_to_rgba_u8, f32_to_rgba_u16).
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
    for (uint32_t p = t; p < npix; p += nrng) {
        const uint32_t x = p % d.w, y = p / d.w;
        const float4 in = src[(size_t)d.astride * (y + FL_GUTTER) + x + FL_GUTTER];   // output.py:28
        T4 o;
        o.x = (T)dclampf(r, (float)PEAK, in.x);
        o.y = (T)dclampf(r, (float)PEAK, in.y);
        o.z = (T)dclampf(r, (float)PEAK, in.z);
        o.w = (T)dclampf(r, (float)PEAK, in.w);
        dst[p] = o;
    }
    rng[t].mul = r.mul; rng[t].state = r.state; rng[t].carry = r.carry;
}

void launch_f32_to_rgba(hipStream_t st, fl_dim d, const float4 *src, fl_mwc *rng, uint32_t nrng, int fmt, void *dst)
{
    dim3 grid((nrng + 255) / 256), block(256);
    if (fmt == 0) hipLaunchKernelGGL((k_f32_to_rgba<uchar4, unsigned char, 255>), grid, block, 0, st, d, src, rng, nrng, (uchar4 *)dst);
    else hipLaunchKernelGGL((k_f32_to_rgba<ushort4, unsigned short, 65535>), grid, block, 0, st, d, src, rng, nrng, (ushort4 *)dst);
}
