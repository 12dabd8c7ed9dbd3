// tone_device.h — per-pixel device functions of the tone-mapping filters, shared by filters.hip (the
// separate kernels) and de.hip (the last DE direction applies them as it stores its result).
#ifndef TONE_DEVICE_H
#define TONE_DEVICE_H
#include "flame_device.h"

// cuburn/code/filters.py:41-53
__device__ __forceinline__ float4 logscale_px(float4 p, float k1, float k2) {
    const float ls = fmaxf(0.0f, fdiv(k1 * flog(1.0f + p.w * k2), p.w));    // NaN at w == 0 -> 0
    p.x *= ls; p.y *= ls; p.z *= ls; p.w *= ls;
    return p;
}

// cuburn/code/filters.py:354-412
__device__ __forceinline__ float4 colorclip_px(float4 p, float vib, float highpow, float gam, float lin, float lingam) {
    if (p.w <= 0.0f) return make_float4(0, 0, 0, 0);
    const float4 o = p;
    float alpha = fpow(p.w, gam);
    if (p.w < lin) {
        const float frac = fdiv(p.w, lin);
        alpha = (1.0f - frac) * p.w * lingam + frac * alpha;
    }
    const float ls = fdiv(vib * alpha, p.w);
    alpha = fminf(1.0f, fmaxf(0.0f, alpha));
    const float maxc = fmaxf(p.x, fmaxf(p.y, p.z));
    const float maxa = maxc * ls;
    const float newls = frcp(maxc);
    if (maxa > 1.0f && highpow >= 0.0f) {
        const float lsratio = fpow(fdiv(newls, ls), highpow);
        p.x *= newls; p.y *= newls; p.z *= newls;
        p.x = maxc - (maxc - p.x) * lsratio;
        p.y = maxc - (maxc - p.y) * lsratio;
        p.z = maxc - (maxc - p.z) * lsratio;
    } else {
        float adjhlp = -highpow;
        if (adjhlp > 1.0f || maxa <= 1.0f) adjhlp = 1.0f;
        if (maxc > 0.0f) {
            const float adj = (1.0f - adjhlp) * newls + adjhlp * ls;
            p.x *= adj; p.y *= adj; p.z *= adj;
        }
    }
    p.x = fminf(1.0f, p.x + (1.0f - vib) * fpow(o.x, gam));
    p.y = fminf(1.0f, p.y + (1.0f - vib) * fpow(o.y, gam));
    p.z = fminf(1.0f, p.z + (1.0f - vib) * fpow(o.z, gam));
    p.w = alpha;
    return p;
}
#endif
