/*
 * filters_ref.c — CPU restatement of cuburn's filter chain and output conversion (oracle).
 * TEST INFRASTRUCTURE ONLY — see flame_ref.h.
 *
 * Follows cuburn/code/filters.py (device code) and cuburn/filters.py (launch order).
 * Images are the padded accumulation buffers: ah rows of astride pixels
 * (cuburn/code/util.py:45-53 launch2 covers the whole padded buffer), float4 = 4
 * interleaved floats.  Texture reads are POINT-sampled with unnormalised coordinates,
 * for which CUDA clamps to the edge whatever address mode was requested
 * (cuburn/code/util.py:55-60; SURVEY.md §7 "Texture semantics").
 *
 * PINNED BY THE REFERENCE'S OWN KERNELS: tests/golden/make_golden_filters.py compiles the assembled CUDA text of
 * cuburn/code/filters.py + color.py as host C++ (a CUDA stand-in, the reference's launch shape and launch order)
 * and tests/test_cpu_golden.py::test_filter_kernels_match_reference_kernels holds every function below to its
 * outputs — 34 kernel outputs, equal to the last bit.
 */
#include "flame_ref.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>
#include <pmmintrin.h>

/* The reference compiles its kernels with -use_fast_math (cuburn/code/util.py:96), which implies
 * -ftz=true: float denormals are read as zero and flushed on output.  That matters in the DE
 * chain, whose weights underflow on sparse images: with denormals kept, `pix.w > 0` holds for a
 * 1e-40 density whose reciprocal is +inf and the colour terms turn into NaN.  Every public entry
 * point below therefore runs with the SSE FTZ and DAZ modes set (and restores the caller's). */
typedef struct { unsigned ftz, daz; } fz_state;
static fz_state fz_enter(void)
{
    fz_state s = { _MM_GET_FLUSH_ZERO_MODE(), _MM_GET_DENORMALS_ZERO_MODE() };
    _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
    _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
    return s;
}
static void fz_leave(fz_state s)
{
    _MM_SET_FLUSH_ZERO_MODE(s.ftz);
    _MM_SET_DENORMALS_ZERO_MODE(s.daz);
}

#define RM_SQRT2 1.41421353816986f

/* cuburn/code/filters.py:8-17 */
static const float patterns[16][2] = {
    {1.0f, 0.0f}, {0.0f, 1.0f}, {1.0f, 1.0f}, {-1.0f, 1.0f},
    {1.0f, 0.5f}, {-0.5f, 1.0f}, {1.0f, -0.5f}, {0.5f, 1.0f},
    {1.0f, 0.666667f}, {-0.666667f, 1.0f}, {1.0f, -0.666667f}, {0.666667f, 1.0f},
    {1.0f, 0.333333f}, {-0.333333f, 1.0f}, {1.0f, -0.333333f}, {0.333333f, 1.0f},
};

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cuburn/code/filters.py:22-35 tex_shear: the offset is rounded to nearest-even BEFORE
 * adding the pixel position */
static inline size_t shear_idx(const ref_dim *d, int pattern, int x, int y, float radius)
{
    float i = rintf(patterns[pattern][0] * radius), j = rintf(patterns[pattern][1] * radius);
    int xi = clampi(x + (int)i, 0, (int)d->astride - 1);
    int yi = clampi(y + (int)j, 0, (int)d->ah - 1);
    return (size_t)yi * d->astride + xi;
}

/* cuburn/code/color.py:25-40 + cuburn/code/filters.py:71-77 */
static void ref_yuv_to_rgb_impl(const ref_dim *d, float *dst, const float *src)
{
    size_t n = (size_t)d->ah * d->astride;
    for (size_t i = 0; i < n; ++i) {
        float Y = src[4 * i], U = src[4 * i + 1], V = src[4 * i + 2], w = src[4 * i + 3];
        U -= 0.5f * w;
        V -= 0.5f * w;
        dst[4 * i] = fmaxf(0.0f, Y + 1.402f * V);
        dst[4 * i + 1] = fmaxf(0.0f, Y - 0.34414f * U - 0.71414f * V);
        dst[4 * i + 2] = fmaxf(0.0f, Y + 1.772f * U);
        dst[4 * i + 3] = w;
    }
}

/* cuburn/code/filters.py:106-117 */
static void ref_den_blur_impl(const ref_dim *d, float *dst, const float *src4, int pattern, int upsample, const float *coefs)
{
    for (int y = 0; y < (int)d->ah; ++y)
        for (int x = 0; x < (int)d->astride; ++x) {
            float den = 0.0f;
            for (int i = 0; i < 7; ++i)
                den += src4[4 * shear_idx(d, pattern, x, y, (float)((i - 3) * (1 << upsample))) + 3] * coefs[i];
            dst[(size_t)y * d->astride + x] = den;
        }
}

/* cuburn/code/filters.py:120-131 */
static void ref_den_blur_1c_impl(const ref_dim *d, float *dst, const float *src1, int pattern, int upsample, const float *coefs)
{
    for (int y = 0; y < (int)d->ah; ++y)
        for (int x = 0; x < (int)d->astride; ++x) {
            float den = 0.0f;
            for (int i = 0; i < 7; ++i)
                den += src1[shear_idx(d, pattern, x, y, (float)((i - 3) * (1 << upsample)))] * coefs[i];
            dst[(size_t)y * d->astride + x] = den;
        }
}

/* cuburn/code/filters.py:136-151 */
static void ref_full_blur_impl(const ref_dim *d, float *dst, const float *src4, int pattern, int upsample, const float *coefs)
{
    for (int y = 0; y < (int)d->ah; ++y)
        for (int x = 0; x < (int)d->astride; ++x) {
            float v[4] = {0, 0, 0, 0};
            for (int i = 0; i < 7; ++i) {
                const float *p = &src4[4 * shear_idx(d, pattern, x, y, (float)((i - 3) * (1 << upsample)))];
                for (int k = 0; k < 4; ++k) v[k] += p[k] * coefs[i];
            }
            memcpy(&dst[4 * ((size_t)y * d->astride + x)], v, sizeof v);
        }
}

/* cuburn/code/filters.py:166-264 */
static void ref_bilateral_impl(const ref_dim *d, float *dst, const float *src4, const float *blur1, int pattern, int radius,
                   float sstd, float cstd, float dstd, float dpow, float gspeed)
{
    float spa[32];
    for (int i = 0; i < 32; ++i) { float df = (float)i; spa[i] = expf(df * df / (-RM_SQRT2 * sstd)); }
    float cscale = 1.0f / (-RM_SQRT2 * 3.0f * cstd);
    float dscale = -0.5f / dstd;
    for (int y = 0; y < (int)d->ah; ++y)
        for (int x = 0; x < (int)d->astride; ++x) {
            size_t gi = (size_t)y * d->astride + x;
            float cen[4];
            memcpy(cen, &src4[4 * gi], sizeof cen);
            float cdrcp = 1.0f / (cen[3] + 1.0e-6f);
            cen[0] *= cdrcp; cen[1] *= cdrcp; cen[2] *= cdrcp;
            float cpowden = powf(cen[3], dpow);
            float out[4] = {0, 0, 0, 0}, weightsum = 0.0f;
            const float *pix = &src4[4 * shear_idx(d, pattern, x, y, (float)(-radius) - 1.0f)];
            const float *next = &src4[4 * shear_idx(d, pattern, x, y, (float)(-radius))];
            for (int r = -radius; r <= radius; ++r) {
                float prev = pix[3];
                pix = next;
                next = &src4[4 * shear_idx(d, pattern, x, y, (float)r + 1.0f)];
                float cdiff = 0.5f;
                if (pix[3] > 0.0f && cen[3] > 0.0f) {
                    float pdrcp = 1.0f / pix[3];
                    float yd = pix[0] * pdrcp - cen[0], ud = pix[1] * pdrcp - cen[1], vd = pix[2] * pdrcp - cen[2];
                    cdiff = yd * yd + ud * ud + vd * vd;
                }
                float powden = powf(pix[3], dpow);
                float dfact = exp2f(dscale * fabsf(cpowden - powden));
                float avg = blur1[shear_idx(d, pattern, x, y, (float)r)];
                float gradfact = (next[3] - prev) / (avg + 1.0e-6f);
                if (r < 0) gradfact = -gradfact;
                gradfact = exp2f(-exp2f(gspeed * gradfact));
                float factor = spa[abs(r)] * expf(cscale * cdiff) * dfact;
                if (r != 0) factor *= gradfact;
                weightsum += factor;
                for (int k = 0; k < 4; ++k) out[k] += factor * pix[k];
            }
            float wr = 1.0f / (weightsum + 1e-10f);
            for (int k = 0; k < 4; ++k) dst[4 * gi + k] = out[k] * wr;
        }
}

/* cuburn/filters.py:62-95 Bilateral.apply: 8 directions of (den_blur -> den_blur_1c ->
 * bilateral r=15), flipping front/back after each.  `front` holds the result on return;
 * `back` (float4) and `side` (>= nbins floats) are scratch. */
static void ref_bilateral_chain_impl(const ref_dim *d, float *front, float *back, float *side, const float *coefs,
                         float sstd, float cstd, float dstd, float dpow, float gspeed)
{
    size_t n = (size_t)d->ah * d->astride;
    float *b1 = malloc(n * sizeof(float));
    float *f = front, *b = back;
    for (int p = 0; p < 8; ++p) {
        ref_den_blur_impl(d, b1, f, p, 0, coefs);
        ref_den_blur_1c_impl(d, side, b1, p, 1, coefs);
        ref_bilateral_impl(d, b, f, side, p, 15, sstd, cstd, dstd, dpow, gspeed);
        float *t = f; f = b; b = t;
    }
    /* 8 flips: result is in the original front */
    free(b1);
}

/* cuburn/code/filters.py:41-53 */
static void ref_logscale_impl(const ref_dim *d, float *buf, float k1, float k2)
{
    size_t n = (size_t)d->ah * d->astride;
    for (size_t i = 0; i < n; ++i) {
        float w = buf[4 * i + 3];
        float ls = fmaxf(0.0f, k1 * logf(1.0f + w * k2) / w);   /* NaN at w == 0 -> 0 via fmaxf */
        for (int k = 0; k < 4; ++k) buf[4 * i + k] *= ls;
    }
}

/* cuburn/code/filters.py:354-412 */
static void ref_colorclip_impl(const ref_dim *d, float *buf, float vib, float highpow, float gam, float lin, float lingam)
{
    size_t n = (size_t)d->ah * d->astride;
    for (size_t i = 0; i < n; ++i) {
        float *p = &buf[4 * i];
        if (p[3] <= 0) { p[0] = p[1] = p[2] = p[3] = 0.0f; continue; }
        float o[3] = {p[0], p[1], p[2]};
        float alpha = powf(p[3], gam);
        if (p[3] < lin) {
            float frac = p[3] / lin;
            alpha = (1.0f - frac) * p[3] * lingam + frac * alpha;
        }
        float ls = vib * alpha / p[3];
        alpha = fminf(1.0f, fmaxf(0.0f, alpha));
        float maxc = fmaxf(p[0], fmaxf(p[1], p[2]));
        float maxa = maxc * ls;
        float newls = 1.0f / maxc;
        if (maxa > 1.0f && highpow >= 0.0f) {
            float lsratio = powf(newls / ls, highpow);
            for (int k = 0; k < 3; ++k) { p[k] *= newls; p[k] = maxc - (maxc - p[k]) * lsratio; }
        } else {
            float adjhlp = -highpow;
            if (adjhlp > 1.0f || maxa <= 1.0f) adjhlp = 1.0f;
            if (maxc > 0.0f) {
                float adj = ((1.0f - adjhlp) * newls + adjhlp * ls);
                for (int k = 0; k < 3; ++k) p[k] *= adj;
            }
        }
        for (int k = 0; k < 3; ++k) {
            p[k] += (1.0f - vib) * powf(o[k], gam);
            p[k] = fminf(1.0f, p[k]);
        }
        p[3] = alpha;
    }
}

/* cuburn/code/filters.py:294-328 + cuburn/filters.py:142-163 */
static void ref_smearclip_chain_impl(const ref_dim *d, float *front, float *back, float *side, const float *coefs,
                         float gam_m_1, float lin, float lingam)
{
    size_t n = (size_t)d->ah * d->astride;
    for (size_t i = 0; i < n; ++i) {       /* apply_gamma_full_hi: front -> side */
        const float *p = &front[4 * i];
        float ls = 0.0f;
        if (p[3] > 0.0f) ls = fmaxf(0.0f, p[3] - 1.0f) / p[3];
        for (int k = 0; k < 4; ++k) side[4 * i + k] = p[k] * ls;
    }
    ref_full_blur_impl(d, back, side, 2, 0, coefs);
    ref_full_blur_impl(d, side, back, 3, 0, coefs);
    ref_full_blur_impl(d, back, side, 0, 0, coefs);
    ref_full_blur_impl(d, side, back, 1, 0, coefs);
    for (size_t i = 0; i < n; ++i) {       /* smearclip: front += side, gamma */
        float *p = &front[4 * i];
        for (int k = 0; k < 4; ++k) p[k] += side[4 * i + k];
        if (p[3] <= 0) { p[0] = p[1] = p[2] = p[3] = 0.0f; continue; }
        float ls = powf(p[3], gam_m_1);
        if (p[3] < lin) {
            float frac = p[3] / lin;
            ls = (1.0f - frac) * lingam + frac * ls;
        }
        for (int k = 0; k < 4; ++k) p[k] *= ls;
    }
}

/* cuburn/code/filters.py:268-288 + cuburn/filters.py:113-130; side/back used as 1-channel scratch */
static void ref_haloclip_chain_impl(const ref_dim *d, float *front, float *back, float *side, const float *coefs, float gam_m_1)
{
    size_t n = (size_t)d->ah * d->astride;
    for (size_t i = 0; i < n; ++i) side[i] = powf(front[4 * i], 0.1f);   /* apply_gamma reads pix.x (:270-271) */
    ref_den_blur_1c_impl(d, back, side, 2, 0, coefs);
    ref_den_blur_1c_impl(d, side, back, 3, 0, coefs);
    for (size_t i = 0; i < n; ++i) {
        float *p = &front[4 * i];
        if (p[3] <= 0) { p[0] = p[1] = p[2] = p[3] = 0.0f; continue; }
        float ls = powf(p[3], gam_m_1) / fmaxf(1.0f, side[i]);
        for (int k = 0; k < 4; ++k) p[k] *= ls;
    }
}

/* cuburn/code/filters.py:332-350 */
static void ref_plainclip_impl(const ref_dim *d, float *buf, float gam_m_1, float lin, float lingam, float brightness)
{
    size_t n = (size_t)d->ah * d->astride;
    for (size_t i = 0; i < n; ++i) {
        float *p = &buf[4 * i];
        if (p[3] <= 0) { p[0] = p[1] = p[2] = p[3] = 0.0f; continue; }
        float ls = powf(p[3], gam_m_1);
        if (p[3] < lin) {
            float frac = p[3] / lin;
            ls = (1.0f - frac) * lingam + frac * ls;
        }
        for (int k = 0; k < 4; ++k) p[k] *= ls * brightness;
    }
}

/* cuburn/code/filters.py:81-90 */
static void ref_logencode_impl(const ref_dim *d, float *dst, const float *src, float degamma)
{
    size_t n = (size_t)d->ah * d->astride * 4;
    for (size_t i = 0; i < n; ++i) dst[i] = log2f(powf(src[i], degamma)) / 12.0f + 1.0f;
}

/* cuburn/code/output.py:7-13 */
static inline float dclampf(ref_mwc *r, float peak, float in)
{
    float ret = 0.0f;
    if (in > 0.0f) ret = fminf(peak, in * peak + 0.99f * ref_mwc_next_01(r));
    return ret;
}

/* cuburn/code/output.py:20-71 f32_to_rgba_u8 / _u16: gutter crop (isrc = sstride*(y+g)+x+g),
 * dithered quantise, truncating convert.  RNG assignment of the device model: state t serves
 * pixels t, t+nrng, t+2*nrng, ... (row-major over the w x h output) in that order. */
static void ref_f32_to_rgba_impl(const ref_dim *d, const float *src, ref_mwc *rng, uint32_t nrng, int fmt, void *dst)
{
    size_t npix = (size_t)d->w * d->h;
    float peak = fmt ? 65535.0f : 255.0f;
    for (uint32_t t = 0; t < nrng; ++t)
        for (size_t p = t; p < npix; p += nrng) {
            uint32_t x = (uint32_t)(p % d->w), y = (uint32_t)(p / d->w);
            const float *in = &src[4 * ((size_t)d->astride * (y + 12) + x + 12)];
            for (int k = 0; k < 4; ++k) {
                float v = dclampf(&rng[t], peak, in[k]);
                if (fmt) ((uint16_t *)dst)[4 * p + k] = (uint16_t)v;
                else ((uint8_t *)dst)[4 * p + k] = (uint8_t)v;
            }
        }
}

/* cuburn/code/output.py:73-236: the planar YUV formats the video encoders take.
 *   fmt 2  f32_to_yuv444p    (:75-102)   u8,  JPEG full-range matrix, three w x h planes
 *   fmt 3  f32_to_yuv444p10  (:106-134)  u16, same matrix, peak 1023; the Cb plane is stored
 *          UNdithered (`dst = 1023.0f * cb`, :129) although its dither draw is made (:122)
 *   fmt 4  f32_to_yuv420p10  (:138-190)  u16, luma per pixel; chroma sample (x, y), x < w/2, y < h/2,
 *          is the alpha-weighted mean over source pixels (2x..2x+1, 2y..2y+1) (:157-183), planes
 *          Y[w*h] Cb[w*h/4] Cr[w*h/4]
 *   fmt 5  f32_to_yuv444p12  (:194-221)  u16, Rec.709 matrix, studio swing (256 + 3504 / 3584), RGB
 *          clamped to [0, 1] first
 * RNG assignment of the device model as for the rgba formats: state t serves pixels t, t+nrng, ...;
 * a pixel draws for Y, Cb, Cr in that order (in 4:2:0 the thread of pixel (x, y) of the top-left
 * quadrant also produces chroma sample (x, y), after its luma, as in the reference). */
static inline uint16_t sat_u16(float v) { return v >= 65535.0f ? 65535 : v > 0.0f ? (uint16_t)v : 0; }

static inline float yuv_cb(const float *in) { return -0.168736f * in[0] - 0.331264f * in[1] + 0.5f * in[2]; }
static inline float yuv_cr(const float *in) { return 0.5f * in[0] - 0.418688f * in[1] - 0.081312f * in[2]; }

static void ref_f32_to_yuv_impl(const ref_dim *d, const float *src, ref_mwc *rng, uint32_t nrng, int fmt, void *dst)
{
    const size_t npix = (size_t)d->w * d->h;
    uint8_t *d8 = (uint8_t *)dst;
    uint16_t *d16 = (uint16_t *)dst;
    for (uint32_t t = 0; t < nrng; ++t)
        for (size_t p = t; p < npix; p += nrng) {
            const uint32_t x = (uint32_t)(p % d->w), y = (uint32_t)(p / d->w);
            const float *in = &src[4 * ((size_t)d->astride * (y + 12) + x + 12)];
            ref_mwc *r = &rng[t];
            if (fmt == 2 || fmt == 3) {
                const float peak = fmt == 2 ? 255.0f : 1023.0f;
                const float cb = yuv_cb(in) + 0.5f;
                const float fy = dclampf(r, peak, 0.299f * in[0] + 0.587f * in[1] + 0.114f * in[2]);
                const float fb = dclampf(r, peak, cb);
                const float fr = dclampf(r, peak, yuv_cr(in) + 0.5f);
                if (fmt == 2) { d8[p] = (uint8_t)fy; d8[npix + p] = (uint8_t)fb; d8[2 * npix + p] = (uint8_t)fr; }
                else { d16[p] = (uint16_t)fy; d16[npix + p] = sat_u16(1023.0f * cb); d16[2 * npix + p] = (uint16_t)fr; }
            } else if (fmt == 4) {
                d16[p] = (uint16_t)dclampf(r, 1023.0f, 0.299f * in[0] + 0.587f * in[1] + 0.114f * in[2]);
                if (x < d->w / 2 && y < d->h / 2) {
                    const float *q = &src[4 * ((size_t)d->astride * (2 * y + 12) + 2 * x + 12)];
                    float sum = (float)((double)q[3] + 1e-12), cb = q[3] * yuv_cb(q), cr = q[3] * yuv_cr(q);
                    const float *q1 = q + 4, *q2 = q + 4 * (size_t)d->astride, *q3 = q2 + 4;
                    sum += q1[3]; cb += q1[3] * yuv_cb(q1); cr += q1[3] * yuv_cr(q1);
                    sum += q2[3]; cb += q2[3] * yuv_cb(q2); cr += q2[3] * yuv_cr(q2);
                    sum += q3[3]; cb += q3[3] * yuv_cb(q3); cr += q3[3] * yuv_cr(q3);
                    const size_t c = (size_t)(d->w / 2) * y + x;
                    d16[npix + c] = (uint16_t)dclampf(r, 1023.0f, cb / sum + 0.5f);
                    d16[npix + npix / 4 + c] = (uint16_t)dclampf(r, 1023.0f, cr / sum + 0.5f);
                }
            } else {
                float c[3];
                for (int k = 0; k < 3; ++k) c[k] = fminf(1.0f, fmaxf(0.0f, in[k]));
                d16[p] = (uint16_t)(dclampf(r, 3504.0f, 0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2]) + 256.0f);
                d16[npix + p] = (uint16_t)(dclampf(r, 3584.0f, -0.11457f * c[0] - 0.38543f * c[1] + 0.5f * c[2] + 0.5f) + 256.0f);
                d16[2 * npix + p] = (uint16_t)(dclampf(r, 3584.0f, 0.5f * c[0] - 0.45416f * c[1] - 0.04585f * c[2] + 0.5f) + 256.0f);
            }
        }
}

/* ---- public entry points: flush-to-zero wrappers (see fz_enter) ---- */
void ref_yuv_to_rgb(const ref_dim *d, float *dst, const float *src)
{
    fz_state s = fz_enter();
    ref_yuv_to_rgb_impl(d, dst, src);
    fz_leave(s);
}
void ref_den_blur(const ref_dim *d, float *dst, const float *src4, int pattern, int upsample, const float *coefs)
{
    fz_state s = fz_enter();
    ref_den_blur_impl(d, dst, src4, pattern, upsample, coefs);
    fz_leave(s);
}
void ref_den_blur_1c(const ref_dim *d, float *dst, const float *src1, int pattern, int upsample, const float *coefs)
{
    fz_state s = fz_enter();
    ref_den_blur_1c_impl(d, dst, src1, pattern, upsample, coefs);
    fz_leave(s);
}
void ref_full_blur(const ref_dim *d, float *dst, const float *src4, int pattern, int upsample, const float *coefs)
{
    fz_state s = fz_enter();
    ref_full_blur_impl(d, dst, src4, pattern, upsample, coefs);
    fz_leave(s);
}
void ref_bilateral(const ref_dim *d, float *dst, const float *src4, const float *blur1, int pattern, int radius,
                   float sstd, float cstd, float dstd, float dpow, float gspeed)
{
    fz_state s = fz_enter();
    ref_bilateral_impl(d, dst, src4, blur1, pattern, radius, sstd, cstd, dstd, dpow, gspeed);
    fz_leave(s);
}
void ref_bilateral_chain(const ref_dim *d, float *front, float *back, float *side, const float *coefs,
                         float sstd, float cstd, float dstd, float dpow, float gspeed)
{
    fz_state s = fz_enter();
    ref_bilateral_chain_impl(d, front, back, side, coefs, sstd, cstd, dstd, dpow, gspeed);
    fz_leave(s);
}
void ref_logscale(const ref_dim *d, float *buf, float k1, float k2)
{
    fz_state s = fz_enter();
    ref_logscale_impl(d, buf, k1, k2);
    fz_leave(s);
}
void ref_colorclip(const ref_dim *d, float *buf, float vib, float highpow, float gam, float lin, float lingam)
{
    fz_state s = fz_enter();
    ref_colorclip_impl(d, buf, vib, highpow, gam, lin, lingam);
    fz_leave(s);
}
void ref_smearclip_chain(const ref_dim *d, float *front, float *back, float *side, const float *coefs,
                         float gam_m_1, float lin, float lingam)
{
    fz_state s = fz_enter();
    ref_smearclip_chain_impl(d, front, back, side, coefs, gam_m_1, lin, lingam);
    fz_leave(s);
}
void ref_haloclip_chain(const ref_dim *d, float *front, float *back, float *side, const float *coefs, float gam_m_1)
{
    fz_state s = fz_enter();
    ref_haloclip_chain_impl(d, front, back, side, coefs, gam_m_1);
    fz_leave(s);
}
void ref_plainclip(const ref_dim *d, float *buf, float gam_m_1, float lin, float lingam, float brightness)
{
    fz_state s = fz_enter();
    ref_plainclip_impl(d, buf, gam_m_1, lin, lingam, brightness);
    fz_leave(s);
}
void ref_logencode(const ref_dim *d, float *dst, const float *src, float degamma)
{
    fz_state s = fz_enter();
    ref_logencode_impl(d, dst, src, degamma);
    fz_leave(s);
}
void ref_f32_to_rgba(const ref_dim *d, const float *src, ref_mwc *rng, uint32_t nrng, int fmt, void *dst)
{
    if (fmt >= 2) ref_f32_to_yuv_impl(d, src, rng, nrng, fmt, dst);
    else ref_f32_to_rgba_impl(d, src, rng, nrng, fmt, dst);      /* integer output: no denormal-sensitive step */
}
