/*
 * flame_ref.c — CPU restatement of cuburn's chaos-game hot path (oracle).
 * TEST INFRASTRUCTURE ONLY — see flame_ref.h.  Scalar C, one walker at a time.
 *
 * Citations are file:line in /root/reference.  The data formats (xform program,
 * parameter block, packed cell) are the documented boundary formats of
 * include/flame_hip.h and are re-derived here independently of the product code.
 *
 * Float arithmetic that must agree bit-for-bit with the device model (affine,
 * colour blend, camera, pixel rounding, flush) is written with explicit fmaf() in a
 * fixed order and compiled with -ffp-contract=off; the device kernel spells the same
 * operations the same way.  Transcendentals use libm here (the device uses fast
 * hardware approximations, as the reference does under -use_fast_math,
 * cuburn/code/util.py:96), so flames using them are compared distributionally.
 *
 * PINNED BY THE REFERENCE'S OWN TEMPLATES (xform application): tests/golden/make_golden_xf.py renders the
 * reference's apply_xf template and its 95 variation entries, compiles them as host C++ and
 * tests/test_cpu_golden.py::test_xform_application_matches_reference_templates holds ref_apply_xf to the results
 * (RNG streams bit-exact, points 1e-4); the packed-cell add with its drain and the flush are held to the reference's inline
 * PTX, interpreted by tests/golden/ptx_mini.py (bit-exact; hot-flag codes 1 and 2 are exchanged there, see
 * test_packed_cell_add_and_flush_match_reference_ptx); parameter blocks and packed palettes to the reference's generated interp
 * kernels.  The control flow around these pieces (xform choice, swap, bounds test) stands on reading: parity of that part is
 * unpinned by reference outputs.
 */
#include "flame_ref.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>

/* float constants of the reference's device prelude, cuburn/code/util.py:148-160 */
#define RM_PI 3.14159274101257f
#define RM_PI_2 1.57079637050629f
#define RM_1_PI 0.31830987334251f
#define RM_2_PI 0.63661974668503f
#define RM_LOG2E 1.44269502162933f
#define RM_SQRT2 1.41421353816986f

/* ------------------------------------------------------------------ RNG */
/* cuburn/code/mwc.py:56-63: t = mul*state + carry; state = lo32, carry = hi32 */
uint32_t ref_mwc_next(ref_mwc *s)
{
    uint64_t t = (uint64_t)s->mul * s->state + s->carry;
    s->state = (uint32_t)t;
    s->carry = (uint32_t)(t >> 32);
    return s->state;
}
/* mwc.py:65-67: u32 -> f32 (round to nearest) times 2^-32; can return exactly 1.0f */
float ref_mwc_next_01(ref_mwc *s) { return (float)ref_mwc_next(s) * (1.0f / 4294967296.0f); }
/* mwc.py:69-77: cvt.rn.f32.s32 then * 2^-31 */
float ref_mwc_next_11(ref_mwc *s) { return (float)(int32_t)ref_mwc_next(s) * (1.0f / 2147483648.0f); }

void ref_mwc_stream(ref_mwc *s, uint32_t n, uint32_t *out)
{
    for (uint32_t i = 0; i < n; ++i) out[i] = ref_mwc_next(s);
}
/* mwc.py:81-88 test_mwc kernel semantics: per-thread u64 sum of `rounds` draws */
void ref_mwc_sums(ref_mwc *s, uint32_t nthreads, uint32_t rounds, uint64_t *sums)
{
    for (uint32_t t = 0; t < nthreads; ++t) {
        uint64_t sum = 0;
        for (uint32_t i = 0; i < rounds; ++i) sum += ref_mwc_next(&s[t]);
        sums[t] = sum;
    }
}

/* ------------------------------------------------------------------ dims */
/* cuburn/render.py:79-89 */
void ref_calc_dim(uint32_t w, uint32_t h, ref_dim *o)
{
    const uint32_t g = 12;
    o->w = w; o->h = h;
    o->aw = w + 2 * g;
    o->ah = 16 * ((h + 2 * g + 15) / 16);
    o->astride = 32 * ((o->aw + 31) / 32);
}

/* ------------------------------------------------------------------ shuffle */
/* Destination thread of the point held by (wave wv, lane l) after the swap of round R.
 * Reference geometry: cuburn/code/iter.py:275-278 (row = ty + tx + (round&1)*(tx/8), same
 * column); helpers/shuf.py:75-80 is the even-round case.  Generalised geometry: nw waves
 * of wl lanes, three phases so that no pair of walkers shares a wave in three consecutive
 * rounds. */
static inline uint32_t shuffle_dest(const ref_geom *g, uint32_t wv, uint32_t l, uint32_t R)
{
    uint32_t nw = (uint32_t)g->nw, wl = (uint32_t)g->wl, sh;
    if (g->ref_shuffle) {
        sh = l + (R & 1) * (l / (nw * wl / 32));
    } else {
        uint32_t ph = R % 3;
        sh = l + (ph == 1 ? l / nw : 0) + (ph == 2 ? l / (nw * nw) : 0);
    }
    return ((wv + sh) % nw) * wl + l;
}
/* out[dst_thread] = src_thread */
void ref_shuffle_perm(const ref_geom *g, uint32_t round, uint32_t *out)
{
    for (int w = 0; w < g->nw; ++w)
        for (int l = 0; l < g->wl; ++l)
            out[shuffle_dest(g, w, l, round)] = w * g->wl + l;
}

/* ------------------------------------------------------------------ splines */
/* cuburn/code/util.py:219-230: rightmost index with hay[idx] strictly below needle, 5 rounds */
static int binsearch32(const float *hay, float needle)
{
    int lo = 0;
    for (int i = 4; i >= 0; --i)
        if (needle > hay[lo + (1 << i)]) lo += 1 << i;
    return lo;
}
#define ELBOW 0.0625f
#define ELOG1 5.0f
static float linlog(float x)   /* interp.py:299-303 */
{
    if (x > ELBOW) return log2f(x) + ELOG1;
    if (x < -ELBOW) return -(log2f(-x) + ELOG1);
    return x / ELBOW;
}
static float linexp(float v)   /* interp.py:306-310 */
{
    if (v >= 1.0f) return exp2f(v - ELOG1);
    if (v <= -1.0f) return -exp2f(-v - ELOG1);
    return v * ELBOW;
}
static float linslope(float x, float m)   /* interp.py:312-316 */
{
    if (x >= ELBOW) return m / x;
    if (x <= -ELBOW) return m / -x;
    return m / ELBOW;
}
/* cuburn/code/interp.py:318-355 */
float ref_catmull_rom(const float *times, const float *knots, float t, int mag)
{
    int idx = binsearch32(times, t);
    if (idx < 1) idx = 1;
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
    if (mag) r = linexp(r);
    return r;
}

/* ------------------------------------------------------------------ palette */
static void rgb2yuv(const float *rgb, float *yuv)   /* cuburn/code/color.py:18-23 */
{
    yuv[0] = 0.299f * rgb[0] + 0.587f * rgb[1] + 0.114f * rgb[2];
    yuv[1] = -0.168736f * rgb[0] - 0.331264f * rgb[1] + 0.5f * rgb[2];
    yuv[2] = 0.5f * rgb[0] - 0.418688f * rgb[1] - 0.081312f * rgb[2];
}
/* float -> u32 as the reference's `uint32_t y = f` (cvt.rzi.u32.f32: truncate, negatives
 * and NaN -> 0, saturating) — interp.py:422-424 */
static uint32_t f2u_trunc(float f)
{
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}
/* cuburn/code/interp.py:372-433.  rng: 64 rows x 256 states, row-major. */
void ref_interp_palette(const float *pal, const float *ptimes, uint32_t npal,
                        float ts, float td, ref_mwc *rng, uint64_t *out)
{
    (void)npal;
    float tstep = td / 64.0f;
    for (int row = 0; row < 64; ++row) {
        float time = ts + (float)row * tstep;
        for (int c = 0; c < 256; ++c) {
            ref_mwc *r = &rng[row * 256 + c];
            int idx = (int)fmaxf((float)(binsearch32(ptimes, time) + 1), 1.0f);
            float tr = ptimes[idx];
            float lf = (tr - time) / (tr - ptimes[idx - 1]);
            float rf = 1.0f - lf;
            const float *left = &pal[((size_t)(idx - 1) * 256 + c) * 4];
            const float *right = &pal[((size_t)idx * 256 + c) * 4];
            if (tr > 1.0f) { right = left; lf = 1.0f; rf = 0.0f; }
            float ly[3], ry[3], yuv[3];
            rgb2yuv(left, ly);
            rgb2yuv(right, ry);
            for (int k = 0; k < 3; ++k) yuv[k] = ly[k] * lf + ry[k] * rf;
            yuv[1] += 0.5f;
            yuv[2] += 0.5f;
            uint32_t y = f2u_trunc(yuv[0] * 255.0f + 0.49f * ref_mwc_next_11(r));
            uint32_t u = f2u_trunc(yuv[1] * 255.0f + 0.49f * ref_mwc_next_11(r));
            uint32_t v = f2u_trunc(yuv[2] * 255.0f + 0.49f * ref_mwc_next_11(r));
            if (y > 255) y = 255;
            if (u > 255) u = 255;
            if (v > 255) v = 255;
            uint32_t hi = (1u << 22) | (y << 4);
            uint32_t lo = (u << 18) | v;
            out[row * 256 + c] = ((uint64_t)hi << 32) | lo;
        }
    }
}

/* ------------------------------------------------------------------ variations */
/* flam3 numbering, cuburn/genome/variations.py:28-127.  v[0] = weight, v[1..] = the
 * variation's genome parameters in sorted-name order, then precalculated values.
 * xf = the owning xform's float record (pre affine xx,xy,xo,yx,yy,yo first).
 * Bodies restate cuburn/code/variations.py:22-988. */
#define W v[0]
#define PA(i) v[1 + (i)]
static int var_apply(int id, const float *v, const float *xf, float *ptx, float *pty,
                     float *pox, float *poy, ref_mwc *r)
{
    float tx = *ptx, ty = *pty, ox = 0.0f, oy = 0.0f;
    float r2 = fmaf(tx, tx, ty * ty);
    switch (id) {
    case 0: /* linear :22 */
        ox = tx * W; oy = ty * W; break;
    case 1: /* sinusoidal :27 */
        ox = W * sinf(tx); oy = W * sinf(ty); break;
    case 2: { /* spherical :32 */
        float k = W / r2; ox = tx * k; oy = ty * k; break; }
    case 3: { /* swirl :38 */
        float c1 = sinf(r2), c2 = cosf(r2);
        ox = W * (c1 * tx - c2 * ty); oy = W * (c2 * tx + c1 * ty); break; }
    case 4: { /* horseshoe :46 */
        float k = W / sqrtf(r2);
        ox = k * (tx - ty) * (tx + ty); oy = 2.0f * tx * ty * k; break; }
    case 5: /* polar :52 */
        ox = W * atan2f(tx, ty) * RM_1_PI; oy = W * (sqrtf(r2) - 1.0f); break;
    case 6: { /* handkerchief :57 */
        float a = atan2f(tx, ty), rr = sqrtf(r2);
        ox = W * rr * sinf(a + rr); oy = W * rr * cosf(a - rr); break; }
    case 7: { /* heart :64 */
        float sq = sqrtf(r2), a = sq * atan2f(tx, ty), rr = W * sq;
        ox = rr * sinf(a); oy = -rr * cosf(a); break; }
    case 8: { /* disc :72 */
        float a = W * atan2f(tx, ty) * RM_1_PI, rr = RM_PI * sqrtf(r2);
        ox = sinf(rr) * a; oy = cosf(rr) * a; break; }
    case 9: { /* spiral :79 */
        float a = atan2f(tx, ty), rr = sqrtf(r2), r1 = W / rr;
        ox = r1 * (cosf(a) + sinf(rr)); oy = r1 * (sinf(a) - cosf(rr)); break; }
    case 10: { /* hyperbolic :87 */
        float a = atan2f(tx, ty), rr = sqrtf(r2);
        ox = W * sinf(a) / rr; oy = W * cosf(a) * rr; break; }
    case 11: { /* diamond :94 */
        float a = atan2f(tx, ty), rr = sqrtf(r2);
        ox = W * sinf(a) * cosf(rr); oy = W * cosf(a) * sinf(rr); break; }
    case 12: { /* ex :101 */
        float a = atan2f(tx, ty), rr = sqrtf(r2);
        float n0 = sinf(a + rr), n1 = cosf(a - rr);
        float m0 = n0 * n0 * n0 * rr, m1 = n1 * n1 * n1 * rr;
        ox = W * (m0 + m1); oy = W * (m0 - m1); break; }
    case 13: { /* julia :112 */
        float a = 0.5f * atan2f(tx, ty);
        if (ref_mwc_next(r) & 1) a += RM_PI;
        float rr = W * sqrtf(sqrtf(r2));
        ox = rr * cosf(a); oy = rr * sinf(a); break; }
    case 14: { /* bent :120 */
        float nx = tx < 0.0f ? 2.0f : 1.0f, ny = ty < 0.0f ? 0.5f : 1.0f;
        ox = W * nx * tx; oy = W * ny * ty; break; }
    case 15: { /* waves :129-140; record: dx2, dy2 */
        float c10 = xf[1], c11 = xf[4];
        ox = W * (tx + c10 * sinf(ty * PA(0))); oy = W * (ty + c11 * sinf(tx * PA(1))); break; }
    case 16: { /* fisheye :142 */
        float k = 2.0f * W / (sqrtf(r2) + 1.0f);
        ox = k * ty; oy = k * tx; break; }
    case 17: { /* popcorn :149 */
        float dx = tanf(3.0f * ty), dy = tanf(3.0f * tx);
        ox = W * (tx + xf[2] * sinf(dx)); oy = W * (ty + xf[5] * sinf(dy)); break; }
    case 18: { /* exponential :156 */
        float dx = W * expf(tx - 1.0f);
        if (isfinite(dx)) { float dy = RM_PI * ty; ox = dx * cosf(dy); oy = dx * sinf(dy); }
        break; }
    case 19: { /* power :165 */
        float a = atan2f(tx, ty), sa = sinf(a), rr = W * powf(sqrtf(r2), sa);
        ox = rr * cosf(a); oy = rr * sa; break; }
    case 20: { /* cosine :173 */
        float a = RM_PI * tx;
        ox = W * cosf(a) * coshf(ty); oy = -W * sinf(a) * sinhf(ty); break; }
    case 21: { /* rings :179 */
        float dx = xf[2]; dx *= dx;
        float rr = sqrtf(r2), a = atan2f(tx, ty);
        rr = W * (fmodf(rr + dx, 2.0f * dx) - dx + rr * (1.0f - dx));
        ox = rr * cosf(a); oy = rr * sinf(a); break; }
    case 22: { /* fan :189 */
        float dx = xf[2]; dx *= dx * RM_PI;
        float dx2 = 0.5f * dx, dy = xf[5], a = atan2f(tx, ty);
        a += (fmodf(a + dy, dx) > dx2) ? -dx2 : dx2;
        float rr = W * sqrtf(r2);
        ox = rr * cosf(a); oy = rr * sinf(a); break; }
    case 23: { /* blob :201; high, low, waves */
        float rr = sqrtf(r2), a = atan2f(tx, ty), bdiff = 0.5f * (PA(0) - PA(1));
        rr *= W * (PA(1) + bdiff * (1.0f + sinf(PA(2) * a)));
        ox = sinf(a) * rr; oy = cosf(a) * rr; break; }
    case 24: { /* pdj :210; a b c d */
        float nx1 = cosf(PA(1) * tx), nx2 = sinf(PA(2) * tx);
        float ny1 = sinf(PA(0) * ty), ny2 = cosf(PA(3) * ty);
        ox = W * (ny1 - nx1); oy = W * (nx2 - ny2); break; }
    case 25: { /* fan2 :219; x y */
        float dy = PA(1), dx = PA(0); dx *= dx * RM_PI;
        float dx2 = 0.5f * dx, a = atan2f(tx, ty), rr = W * sqrtf(r2);
        float t = a + dy - dx * truncf((a + dy) / dx);
        if (t > dx2) a -= dx2; else a += dx2;
        ox = rr * sinf(a); oy = rr * cosf(a); break; }
    case 26: { /* rings2 :236; val */
        float dx = PA(0); dx *= dx;
        float rr = sqrtf(r2), a = atan2f(tx, ty);
        rr += -2.0f * dx * (float)(int)((rr + dx) / (2.0f * dx)) + rr * (1.0f - dx);
        ox = W * sinf(a) * rr; oy = W * cosf(a) * rr; break; }
    case 27: { /* eyefish :246 */
        float k = 2.0f * W / (sqrtf(r2) + 1.0f); ox = k * tx; oy = k * ty; break; }
    case 28: { /* bubble :252 */
        float k = W / (0.25f * r2 + 1.0f); ox = k * tx; oy = k * ty; break; }
    case 29: /* cylinder :258 */
        ox = W * sinf(tx); oy = W * ty; break;
    case 30: { /* perspective :263-273; angle dist | mdist sin cos */
        float t = 1.0f / (PA(2) - ty * PA(3));
        ox = W * PA(2) * tx * t; oy = W * PA(4) * ty * t; break; }
    case 31: { /* noise :275 */
        float tmpr = ref_mwc_next_01(r) * 2.0f * RM_PI, rr = W * ref_mwc_next_01(r);
        ox = tx * rr * cosf(tmpr); oy = ty * rr * sinf(tmpr); break; }
    case 32: { /* julian :282-294; dist power | cn */
        float power = PA(1);
        float t_rnd = truncf(ref_mwc_next_01(r) * fabsf(power));
        float a = atan2f(ty, tx);
        float tmpr = (a + 2.0f * RM_PI * t_rnd) / power;
        float rr = W * powf(r2, PA(2));
        ox = rr * cosf(tmpr); oy = rr * sinf(tmpr); break; }
    case 33: { /* juliascope :296-309; dist power | cn */
        float ang = atan2f(ty, tx), power = PA(1);
        float t_rnd = truncf(ref_mwc_next_01(r) * fabsf(power));
        if (ref_mwc_next(r) & 1) ang = -ang;
        float tmpr = (2.0f * RM_PI * t_rnd + ang) / power;
        float rr = W * powf(r2, PA(2));
        ox = rr * cosf(tmpr); oy = rr * sinf(tmpr); break; }
    case 34: { /* blur :311 */
        float tmpr = ref_mwc_next_01(r) * 2.0f * RM_PI, rr = W * ref_mwc_next_01(r);
        ox = rr * cosf(tmpr); oy = rr * sinf(tmpr); break; }
    case 35: { /* gaussian_blur :318 */
        float ang = ref_mwc_next_01(r) * 2.0f * RM_PI;
        float rr = W * 0.57736f * sqrtf(-2.0f * log2f(ref_mwc_next_01(r)) / RM_LOG2E);
        ox = rr * cosf(ang); oy = rr * sinf(ang); break; }
    case 36: { /* radial_blur :328; angle */
        float ba = PA(0) * RM_PI * 0.5f, spinvar = sinf(ba), zoomvar = cosf(ba);
        float rr = W * 0.57736f * sqrtf(-2.0f * log2f(ref_mwc_next_01(r)) / RM_LOG2E);
        float ra = sqrtf(r2), tmpa = atan2f(ty, tx) + spinvar * rr, rz = zoomvar * rr - 1.0f;
        ox = ra * cosf(tmpa) + rz * tx; oy = ra * sinf(tmpa) + rz * ty; break; }
    case 37: { /* pie :341; rotation slices thickness */
        float slices = PA(1);
        float sl = truncf(ref_mwc_next_01(r) * slices + 0.5f);
        float a = PA(0) + 2.0f * RM_PI * (sl + ref_mwc_next_01(r) * PA(2)) / slices;
        float rr = W * ref_mwc_next_01(r);
        ox = rr * cosf(a); oy = rr * sinf(a); break; }
    case 38: { /* ngon :351; circle corners power sides */
        float power = PA(2) * 0.5f, b = 2.0f * RM_PI / PA(3);
        float r_factor = powf(r2, power), theta = atan2f(ty, tx);
        float phi = theta - b * floorf(theta / b);
        if (phi > b / 2.0f) phi -= b;
        float amp = (PA(1) * (1.0f / cosf(phi) - 1.0f) + PA(0)) / r_factor;
        ox = W * tx * amp; oy = W * ty * amp; break; }
    case 39: { /* curl :367; c1 c2 */
        float c1 = PA(0), c2 = PA(1);
        float re = 1.0f + c1 * tx + c2 * (tx * tx - ty * ty), im = c1 * ty + 2.0f * c2 * tx * ty;
        float k = W / (re * re + im * im);
        ox = k * (tx * re + ty * im); oy = k * (ty * re - tx * im); break; }
    case 40: { /* rectangles :379; x y */
        float rx = PA(0), ry = PA(1);
        ox = W * ((rx == 0.0f) ? tx : rx * (2.0f * floorf(tx / rx) + 1.0f) - tx);
        oy = W * ((ry == 0.0f) ? ty : ry * (2.0f * floorf(ty / ry) + 1.0f) - ty); break; }
    case 41: { /* arch :387 */
        float ang = ref_mwc_next_01(r) * W * RM_PI;
        ox = W * sinf(ang); oy = W * sinf(ang) * sinf(ang) / cosf(ang); break; }
    case 42: /* tangent :394 */
        ox = W * sinf(tx) / cosf(ty); oy = W * tanf(ty); break;
    case 43: /* square :399 */
        ox = W * (ref_mwc_next_01(r) - 0.5f); oy = W * (ref_mwc_next_01(r) - 0.5f); break;
    case 44: { /* rays :404 */
        float ang = W * ref_mwc_next_01(r) * RM_PI, k = W / r2, tanr = W * tanf(ang) * k;
        ox = tanr * cosf(tx); oy = tanr * sinf(ty); break; }
    case 45: { /* blade :412 */
        float rr = ref_mwc_next_01(r) * W * sqrtf(r2);
        ox = W * tx * (cosf(rr) + sinf(rr)); oy = W * tx * (cosf(rr) - sinf(rr)); break; }
    case 46: { /* secant2 :418 */
        float rr = W * sqrtf(r2), cr = cosf(rr), icr = 1.0f / cr;
        icr += (cr < 0 ? 1.0f : -1.0f);
        ox = W * tx; oy = W * icr; break; }
    case 48: { /* cross :430 */
        float s = tx * tx - ty * ty, k = W * sqrtf(1.0f / (s * s));
        ox = k * tx; oy = k * ty; break; }
    case 49: { /* disc2 :438; rot twist */
        float twist = PA(1), rotpi = PA(0) * RM_PI;
        float sintwist = sinf(twist), costwist = cosf(twist) - 1.0f;
        if (twist > 2.0f * RM_PI) { float k = (1.0f + twist - 2.0f * RM_PI); sintwist *= k; costwist *= k; }
        if (twist < -2.0f * RM_PI) { float k = (1.0f + twist + 2.0f * RM_PI); sintwist *= k; costwist *= k; }
        float t = rotpi * (tx + ty), k = W * atan2f(tx, ty) / RM_PI;
        ox = k * (sinf(t) + costwist); oy = k * (cosf(t) + sintwist); break; }
    case 50: { /* super_shape :464; holes m n1 n2 n3 rnd */
        float ang = atan2f(ty, tx), theta = 0.25f * (PA(1) * ang + RM_PI);
        float t1 = powf(fabsf(cosf(theta)), PA(3)), t2 = powf(fabsf(sinf(theta)), PA(4));
        float myrnd = PA(5), d = sqrtf(r2);
        float k = W * ((myrnd * ref_mwc_next_01(r) + (1.0f - myrnd) * d) - PA(0))
                * powf(t1 + t2, -1.0f / PA(2)) / d;
        ox = k * tx; oy = k * ty; break; }
    case 51: { /* flower :482; holes petals */
        float k = W * (ref_mwc_next_01(r) - PA(0)) * cosf(PA(1) * atan2f(ty, tx)) / sqrtf(r2);
        ox = k * tx; oy = k * ty; break; }
    case 52: { /* conic :493; eccentricity holes */
        float d = sqrtf(r2), ct = tx / d;
        float k = W * (ref_mwc_next_01(r) - PA(1)) * PA(0) / (1.0f + PA(0) * ct) / d;
        ox = k * tx; oy = k * ty; break; }
    case 53: { /* parabola :505; height width */
        float rr = sqrtf(r2), sr = sinf(rr), cr = cosf(rr);
        ox = PA(0) * W * sr * sr * ref_mwc_next_01(r);
        oy = PA(1) * W * cr * ref_mwc_next_01(r); break; }
    case 54: { /* bent2 :514; x y */
        float nx = tx < 0.0f ? PA(0) : 1.0f, ny = ty < 0.0f ? PA(1) : 1.0f;
        ox = W * nx * tx; oy = W * ny * ty; break; }
    case 55: { /* bipolar :523; shift */
        float t = r2 + 1.0f, x2 = tx * 2.0f, ps = -RM_PI_2 * PA(0);
        float y = 0.5f * atan2f(2.0f * ty, r2 - 1.0f) + ps;
        if (y > RM_PI_2) y = -RM_PI_2 + fmodf(y + RM_PI_2, RM_PI);
        else if (y < -RM_PI_2) y = RM_PI_2 - fmodf(RM_PI_2 - y, RM_PI);
        ox = W * 0.25f * RM_2_PI * logf((t + x2) / (t - x2)); oy = W * RM_2_PI * y; break; }
    case 56: { /* boarders :539 */
        float rx = rintf(tx), ry = rintf(ty), fx = tx - rx, fy = ty - ry;
        if (ref_mwc_next_01(r) > 0.75f) {
            ox = W * (fx * 0.5f + rx); oy = W * (fy * 0.5f + ry);
        } else if (fabsf(fx) >= fabsf(fy)) {
            if (fx >= 0.0f) { ox = W * (fx * 0.5f + rx + 0.25f); oy = W * (fy * 0.5f + ry + 0.25f * fy / fx); }
            else            { ox = W * (fx * 0.5f + rx - 0.25f); oy = W * (fy * 0.5f + ry - 0.25f * fy / fx); }
        } else {
            if (fy >= 0.0f) { oy = W * (fy * 0.5f + ry + 0.25f); ox = W * (fx * 0.5f + rx + fx / fy * 0.25f); }
            else            { oy = W * (fy * 0.5f + ry - 0.25f); ox = W * (fx * 0.5f + rx - fx / fy * 0.25f); }
        }
        break; }
    case 57: { /* butterfly :571 */
        float wx = W * 1.3029400317411197908970256609023f, y2 = ty * 2.0f;
        float k = wx * sqrtf(fabsf(ty * tx) / (tx * tx + y2 * y2));
        ox = k * tx; oy = k * y2; break; }
    case 58: { /* cell :580; size */
        float cs = PA(0), ics = 1.0f / cs;
        float cx = floorf(tx * ics), cy = floorf(ty * ics);
        float dx = tx - cx * cs, dy = ty - cy * cs;
        if (cy >= 0.0f) { if (cx >= 0.0f) { cy *= 2.0f; cx *= 2.0f; } else { cy *= 2.0f; cx = -(2.0f * cx + 1.0f); } }
        else { if (cx >= 0.0f) { cy = -(2.0f * cy + 1.0f); cx *= 2.0f; } else { cy = -(2.0f * cy + 1.0f); cx = -(2.0f * cx + 1.0f); } }
        ox = W * (dx + cx * cs); oy = -W * (dy + cy * cs); break; }
    case 59: { /* cpow :612; i power r */
        float a = atan2f(ty, tx), lnr = 0.5f * logf(r2), power = 1.0f / PA(1);
        float va = 2.0f * RM_PI * power, vc = PA(2) * power, vd = PA(0) * power;
        float ang = vc * a + vd * lnr + va * floorf(power * ref_mwc_next_01(r));
        float m = W * expf(vc * lnr - vd * a);
        ox = m * cosf(ang); oy = m * sinf(ang); break; }
    case 60: { /* curve :625-634; xamp xlength yamp ylength | x2 y2 */
        ox = W * (tx + PA(0) * expf(-ty * ty * PA(4))); oy = W * (ty + PA(2) * expf(-tx * tx * PA(5))); break; }
    case 61: { /* edisc :636 */
        float tmp = r2 + 1.0f, tmp2 = 2.0f * tx;
        float r1 = sqrtf(tmp + tmp2), rr2 = sqrtf(tmp - tmp2), xmax = (r1 + rr2) * 0.5f;
        float a1 = logf(xmax + sqrtf(xmax - 1.0f)), a2 = -acosf(tx / xmax), nw = W / 11.57034632f;
        float snv = sinf(a1), csv = cosf(a1);
        if (ty > 0.0f) snv = -snv;
        ox = nw * coshf(a2) * csv; oy = nw * sinhf(a2) * snv; break; }
    case 62: { /* elliptic :654 */
        float tmp = r2 + 1.0f, x2 = 2.0f * tx, xmax = 0.5f * (sqrtf(tmp + x2) + sqrtf(tmp - x2));
        float a = tx / xmax, b = 1.0f - a * a, ssx = xmax - 1.0f, nw = W / RM_PI_2;
        b = b < 0.0f ? 0.0f : sqrtf(b);
        ssx = ssx < 0.0f ? 0.0f : sqrtf(ssx);
        ox = nw * atan2f(a, b);
        oy = ty > 0.0f ? nw * logf(xmax + ssx) : -nw * logf(xmax + ssx); break; }
    case 63: { /* escher :682; beta */
        float a = atan2f(ty, tx), lnr = 0.5f * logf(r2), seb = sinf(PA(0)), ceb = cosf(PA(0));
        float vc = 0.5f * (1.0f + ceb), vd = 0.5f * seb;
        float m = W * expf(vc * lnr - vd * a), n = vc * a + vd * lnr;
        ox = m * cosf(n); oy = m * sinf(n); break; }
    case 64: { /* foci :697 */
        float expx = expf(tx) * 0.5f, expnx = 0.25f / expx, sn = sinf(ty), cn = cosf(ty);
        float tmp = W / (expx + expnx - cn);
        ox = tmp * (expx - expnx); oy = tmp * sn; break; }
    case 65: { /* lazysusan :707; space spin twist x y */
        float lx = PA(3), ly = PA(4), x = tx - lx, y = ty + ly, rr = sqrtf(x * x + y * y);
        if (rr < W) {
            float a = atan2f(y, x) + PA(1) + PA(2) * (W - rr);
            ox = W * (rr * cosf(a) + lx); oy = W * (rr * sinf(a) - ly);
        } else {
            rr = 1.0f + PA(0) / rr;
            ox = W * (rr * x + lx); oy = W * (rr * y - ly);
        }
        break; }
    case 66: { /* loonie :728 */
        float w2 = W * W;
        if (r2 < w2) { float k = W * sqrtf(w2 / r2 - 1.0f); ox = k * tx; oy = k * ty; }
        else { ox = W * tx; oy = W * ty; }
        break; }
    case 67: { /* pre_blur :741-749: mutates tx, ty for the variations that follow */
        float rndG = W * (ref_mwc_next_01(r) + ref_mwc_next_01(r) + ref_mwc_next_01(r)
                          + ref_mwc_next_01(r) - 2.0f);
        float rndA = ref_mwc_next_01(r) * 2.0f * RM_PI;
        *ptx = tx + rndG * cosf(rndA); *pty = ty + rndG * sinf(rndA); break; }
    case 68: { /* modulus :751; x y */
        float mx = PA(0), my = PA(1), xr = 2.0f * mx, yr = 2.0f * my;
        if (tx > mx) ox = W * (-mx + fmodf(tx + mx, xr));
        else if (tx < -mx) ox = W * (mx - fmodf(mx - tx, xr));
        else ox = W * tx;
        if (ty > my) oy = W * (-my + fmodf(ty + my, yr));
        else if (ty < -my) oy = W * (my - fmodf(my - ty, yr));
        else oy = W * ty;
        break; }
    case 69: { /* oscope :771; amplitude damping frequency separation */
        float tpf = 2.0f * RM_PI * PA(2);
        float t = PA(0) * expf(-fabsf(tx) * PA(1)) * cosf(tpf * tx) + PA(3);
        ox = W * tx; oy = (fabsf(ty) <= t) ? -W * ty : W * ty; break; }
    case 70: { /* polar2 :786 */
        float p2v = W / RM_PI; ox = p2v * atan2f(tx, ty); oy = 0.5f * p2v * logf(r2); break; }
    case 71: { /* popcorn2 :792; c x y */
        ox = W * (tx + PA(1) * sinf(tanf(ty * PA(0)))); oy = W * (ty + PA(2) * sinf(tanf(tx * PA(0)))); break; }
    case 72: { /* scry :798 */
        float k = 1.0f / (sqrtf(r2) * (r2 + 1.0f / W)); ox = tx * k; oy = ty * k; break; }
    case 73: { /* separation :808; x xinside y yinside */
        float sx2 = PA(0) * PA(0), sy2 = PA(2) * PA(2);
        ox = tx > 0.0f ? W * (sqrtf(tx * tx + sx2) - tx * PA(1)) : -W * (sqrtf(tx * tx + sx2) + tx * PA(1));
        oy = ty > 0.0f ? W * (sqrtf(ty * ty + sy2) - ty * PA(3)) : -W * (sqrtf(ty * ty + sy2) + ty * PA(3));
        break; }
    case 74: { /* split :823; xsize ysize */
        oy = (cosf(tx * PA(0) * RM_PI) >= 0.0f) ? W * ty : -W * ty;
        ox = (cosf(ty * PA(1) * RM_PI) >= 0.0f) ? W * tx : -W * tx; break; }
    case 75: /* splits :835; x y */
        ox = W * (tx + copysignf(PA(0), tx)); oy = W * (ty + copysignf(PA(1), ty)); break;
    case 76: { /* stripes :840; space warp */
        float roundx = floorf(tx + 0.5f), offsetx = tx - roundx;
        ox = W * (offsetx * (1.0f - PA(0)) + roundx); oy = W * (ty + offsetx * offsetx * PA(1)); break; }
    case 77: { /* wedge :847; angle count hole swirl */
        float rr = sqrtf(r2), a = atan2f(ty, tx) + PA(3) * rr, wc = PA(1), wa = PA(0);
        float c = floorf((wc * a + RM_PI) * RM_1_PI * 0.5f);
        float comp_fac = 1.0f - wa * wc * RM_1_PI * 0.5f;
        a = a * comp_fac + c * wa;
        rr = W * (rr + PA(2));
        ox = rr * cosf(a); oy = rr * sinf(a); break; }
    case 80: { /* whorl :860; inside outside */
        float rr = sqrtf(r2), a = atan2f(ty, tx);
        a += (rr < W ? PA(0) : PA(1)) / (W - rr);
        ox = W * rr * cosf(a); oy = W * rr * sinf(a); break; }
    case 81: /* waves2 :873; freqx freqy scalex scaley */
        ox = W * (tx + PA(2) * sinf(ty * PA(0))); oy = W * (ty + PA(3) * sinf(tx * PA(1))); break;
    case 82: { /* exp :878 */
        float e = expf(tx); ox = W * e * cosf(ty); oy = W * e * sinf(ty); break; }
    case 83: /* log :884 */
        ox = W * 0.5f * logf(r2); oy = W * atan2f(ty, tx); break;
    case 84: /* sin :889 */
        ox = W * sinf(tx) * coshf(ty); oy = W * cosf(tx) * sinhf(ty); break;
    case 85: /* cos :894 */
        ox = W * cosf(tx) * coshf(ty); oy = -W * sinf(tx) * sinhf(ty); break;
    case 86: { /* tan :899 */
        float d = 1.0f / (cosf(2.0f * tx) + coshf(2.0f * ty));
        ox = W * d * sinf(2.0f * tx); oy = W * d * sinhf(2.0f * ty); break; }
    case 87: { /* sec :905 */
        float d = 2.0f / (cosf(2.0f * tx) + coshf(2.0f * ty));
        ox = W * d * cosf(tx) * coshf(ty); oy = W * d * sinf(tx) * sinhf(ty); break; }
    case 88: { /* csc :911 */
        float d = 2.0f / (coshf(2.0f * ty) - cosf(2.0f * tx));
        ox = W * d * sinf(tx) * coshf(ty); oy = -W * d * cosf(tx) * sinhf(ty); break; }
    case 89: { /* cot :917 */
        float d = 1.0f / (coshf(2.0f * ty) - cosf(2.0f * tx));
        ox = W * d * sinf(2.0f * tx); oy = W * d * -1.0f * sinhf(2.0f * ty); break; }
    case 90: /* sinh :923 */
        ox = W * sinhf(tx) * cosf(ty); oy = W * coshf(tx) * sinf(ty); break;
    case 91: /* cosh :928 */
        ox = W * coshf(tx) * cosf(ty); oy = W * sinhf(tx) * sinf(ty); break;
    case 92: { /* tanh :933 */
        float d = 1.0f / (cosf(2.0f * ty) + coshf(2.0f * tx));
        ox = W * d * sinhf(2.0f * tx); oy = W * d * sinf(2.0f * ty); break; }
    case 93: { /* sech :939 */
        float d = 2.0f / (cosf(2.0f * ty) + coshf(2.0f * tx));
        ox = W * d * cosf(ty) * coshf(tx); oy = -W * d * sinf(ty) * sinhf(tx); break; }
    case 94: { /* csch :945 */
        float d = 2.0f / (coshf(2.0f * tx) - cosf(2.0f * ty));
        ox = W * d * sinhf(tx) * cosf(ty); oy = -W * d * coshf(tx) * sinf(ty); break; }
    case 95: { /* coth :951 */
        float d = 1.0f / (coshf(2.0f * tx) - cosf(2.0f * ty));
        ox = W * d * sinhf(2.0f * tx); oy = W * d * sinf(2.0f * ty); break; }
    case 97: { /* flux :957; spread */
        float xpw = tx + W, xmw = tx - W;
        float avgr = W * (2.0f + PA(0)) * sqrtf(sqrtf(ty * ty + xpw * xpw) / sqrtf(ty * ty + xmw * xmw));
        float avga = (atan2f(ty, xmw) - atan2f(ty, xpw)) * 0.5f;
        ox = avgr * cosf(avga); oy = avgr * sinf(avga); break; }
    case 98: { /* mobius :967; im_a im_b im_c im_d re_a re_b re_c re_d */
        float ima = PA(0), imb = PA(1), imc = PA(2), imd = PA(3);
        float rea = PA(4), reb = PA(5), rec = PA(6), red = PA(7);
        float re_u = rea * tx - ima * ty + reb, im_u = rea * ty + ima * tx + imb;
        float re_v = rec * tx - imc * ty + red, im_v = rec * ty + imc * tx + imd;
        float rad_v = W / (re_v * re_v + im_v * im_v);
        ox = rad_v * (re_u * re_v + im_u * im_v); oy = rad_v * (im_u * re_v - re_u * im_v); break; }
    default:
        return -1;
    }
    *pox += ox;
    *poy += oy;
    return 0;
}
#undef W
#undef PA

int ref_var_supported(int id)
{
    float v[16] = {0}, xf[16] = {0}, tx = 0.3f, ty = 0.2f, ox = 0, oy = 0;
    ref_mwc r = {4294967118u, 1, 1};
    return var_apply(id, v, xf, &tx, &ty, &ox, &oy, &r) == 0;
}

/* cuburn/code/iter.py:121-149: pre affine -> sum of variations -> optional post affine ->
 * colour blend.  Affine / blend spelled with explicit fmaf (device model agreement).
 * Record layout: include/flame_hip.h (5). */
int ref_apply_xf(const int32_t *prog, const float *P, int xfi, float *px, float *py, float *pc, ref_mwc *r)
{
    const float *xf = P + prog[5] + (size_t)xfi * prog[6];
    const int vstride = prog[7];
    int32_t word14;
    memcpy(&word14, &xf[14], 4);
    int nvar = word14 & 0xff, post = (word14 >> 8) & 1;
    float x = *px, y = *py;
    float tx = fmaf(xf[0], x, fmaf(xf[1], y, xf[2]));
    float ty = fmaf(xf[3], x, fmaf(xf[4], y, xf[5]));
    float ox = -0.0f, oy = -0.0f;        /* as the device: -0 + v = v exactly, so the sum is the variations' own (a sum of +0 terms only differs in the sign of zero) */
    for (int j = 0; j < nvar; ++j) {
        const float *v = xf + 16 + j * vstride;
        int32_t vid;
        memcpy(&vid, &v[0], 4);
        if (var_apply(vid, v + 1, xf, &tx, &ty, &ox, &oy, r)) return -1;
    }
    if (post) {
        const float *q = xf + 6;
        float qx = fmaf(q[0], ox, fmaf(q[1], oy, q[2]));
        float qy = fmaf(q[3], ox, fmaf(q[4], oy, q[5]));
        ox = qx; oy = qy;
    }
    float csp = xf[13];
    *pc = fmaf(*pc, 1.0f - csp, xf[12] * csp);
    *px = ox;
    *py = oy;
    return 0;
}

int ref_apply_xf_n(const int32_t *prog, const float *P, int xfi, uint32_t n, float *xyzw, ref_mwc *r)
{
    for (uint32_t i = 0; i < n; ++i) {
        int rc = ref_apply_xf(prog, P, xfi, &xyzw[4 * i], &xyzw[4 * i + 1], &xyzw[4 * i + 2], &r[i]);
        if (rc) return rc;
    }
    return 0;
}

/* cuburn/code/iter.py:260-272: first xform whose cumulative density is >= the selector */
static inline int select_xf(const int32_t *prog, const float *P, float sel)
{
    int nxf = prog[1];
    const float *cdf = P + prog[4];
    for (int i = 0; i < nxf - 1; ++i)
        if (sel <= cdf[i]) return i;
    return nxf - 1;
}

/* cuburn/code/util.py:194-200 trunca = cvt.rni.s32.f32 (round to nearest even, saturating),
 * result reinterpreted as unsigned.  PTX maps NaN to 0, which would plot a NaN point at column
 * / row 0 of the gutter; here NaN is mapped out of range (the sample is dropped). */
static inline uint32_t trunca(float f)
{
    if (f != f) return 0x80000000u;
    if (f >= 2147483648.0f) return 0x7fffffffu;
    if (f <= -2147483648.0f) return 0x80000000u;
    return (uint32_t)(int32_t)rintf(f);
}

/* cuburn/code/iter.py:385-389,464-468 */
void ref_unpack_cell(uint64_t cell, uint32_t o[4])
{
    uint32_t hi = (uint32_t)(cell >> 32), lo = (uint32_t)cell;
    o[3] = hi >> 22;                                 /* count */
    o[0] = (hi >> 4) & 0x3ffff;                      /* sum Y */
    o[1] = ((hi & 0xf) << 14) | (lo >> 18);          /* sum U */
    o[2] = lo & 0x3ffff;                             /* sum V */
}

#define INV255 0.003921568859368562698f
/* hot flag -> multiplier, cuburn/code/iter.py:326 / :447-452: (1 << (2*flag)) >> 1, 0 -> 1 */
static inline float hot_mult(uint32_t flag) { return flag ? (float)((1u << (flag << 1)) >> 1) : 1.0f; }

static void spill_cell(uint64_t cell, float mult, float *o4)   /* iter.py:383-405 */
{
    uint32_t u[4];
    ref_unpack_cell(cell, u);
    float m255 = mult * INV255;
    o4[0] += (float)u[0] * m255;
    o4[1] += (float)u[1] * m255;
    o4[2] += (float)u[2] * m255;
    o4[3] += (float)u[3] * mult;
}

/* One sample into the packed histogram (cuburn/code/iter.py:332-411, the checked path on every add): the palette cell of the
 * dithered colour index is added to the pixel's packed cell; a cell seen at 512 hits or more is taken out whole and added to
 * the float accumulator.  Returns 1 if it drained the cell. */
static inline int plot_cell(const uint64_t *palrow, uint32_t gi, float cc, float dither, float mult, uint64_t *atom, float *out4)
{
    float cf = fmaf(cc, 255.0f, dither);                             /* iter.py:346-348 */
    int ci = (cf != cf) ? 0 : (cf >= 255.0f ? 255 : (cf <= 0.0f ? 0 : (int)rintf(cf)));
    uint64_t val = palrow[ci];                                       /* iter.py:351 (clamped surface read) */
    uint64_t old = atom[gi];
    atom[gi] = old + val;                                            /* iter.py:355-363 */
    if ((uint32_t)(old >> 32) >= (256u << 23)) {                     /* iter.py:369-379 checked path */
        spill_cell(atom[gi], mult, &out4[4 * (size_t)gi]);
        atom[gi] = 0;
        return 1;
    }
    return 0;
}

/* Test hook: plot_cell on explicit samples, in order (tests/test_cpu_golden.py holds it to the reference's own PTX run through
 * tests/golden/ptx_mini.py).  `palette` is [64][256] packed cells, `row[i]` the palette row (temporal sample) of sample i. */
void ref_plot_samples(uint32_t n, const uint32_t *gi, const float *cc, const float *dither, const uint32_t *row,
                      const float *mult, const uint64_t *palette, uint64_t *atom, float *out4)
{
    for (uint32_t i = 0; i < n; ++i)
        plot_cell(palette + (size_t)row[i] * 256, gi[i], cc[i], dither[i], mult[i], atom, out4);
}

/* One workgroup of the device model (cuburn/code/iter.py:157-418, adapted as documented in
 * DESIGN.md §iterate): walkers are bound to their slot, the per-wave selector is lane 0's
 * draw, fuse is a property of the launch, hot flags are 2 bits/pixel at word gi>>4. */
static int iter_block(const ref_geom *g, const ref_dim *dim, const int32_t *prog, const float *P,
                      const uint64_t *palrow, ref_mwc *rng, float *pts,
                      const uint32_t *hot, uint64_t *atom, float *out4,
                      uint32_t round0, uint32_t nrounds, uint32_t fuse, uint64_t ctr[4])
{
    const int nt = g->nw * g->wl;
    int has_final = prog[2], nxf = prog[1];
    float *sx = malloc(sizeof(float) * nt * 3), *sy = sx + nt, *sc = sy + nt;
    float *dith = malloc(sizeof(float) * nt);
    uint32_t *sel = malloc(sizeof(uint32_t) * nt);
    uint32_t *sel_next = malloc(sizeof(uint32_t) * nt);
    for (int t = 0; t < nt; ++t) {
        dith[t] = 0.49f * ref_mwc_next_11(&rng[t]);                 /* iter.py:185 */
        float x = pts[4 * t], y = pts[4 * t + 1];
        if (!isfinite(fabsf(x) + fabsf(y))) {                       /* iter.py:209-216 */
            pts[4 * t] = ref_mwc_next_11(&rng[t]);
            pts[4 * t + 1] = ref_mwc_next_11(&rng[t]);
            pts[4 * t + 2] = ref_mwc_next_01(&rng[t]);
        }
        sel_next[t] = ref_mwc_next(&rng[t]);                        /* selector of round 0 */
    }
    for (uint32_t rd = 0; rd < nrounds; ++rd) {
        uint32_t R = round0 + rd;
        for (int t = 0; t < nt; ++t) {
            float x = pts[4 * t], y = pts[4 * t + 1];
            if (!isfinite(fabsf(x) + fabsf(y))) {                   /* iter.py:225-229 */
                pts[4 * t] = ref_mwc_next_11(&rng[t]);
                pts[4 * t + 1] = ref_mwc_next_11(&rng[t]);
                pts[4 * t + 2] = ref_mwc_next_01(&rng[t]);
            }
            /* device model: the selector of round r+1 is drawn at the top of round r (so the
             * kernel can fetch the chosen record a round ahead); the last one goes unused */
            sel[t] = sel_next[t];
            sel_next[t] = ref_mwc_next(&rng[t]);
        }
        for (int t = 0; t < nt; ++t) {
            int wv = t / g->wl, l = t % g->wl;
            uint32_t s = sel[wv * g->wl];                            /* wave-uniform selector */
            float xfsel = (float)s * (1.0f / 4294967296.0f);
            int k = select_xf(prog, P, xfsel);
            float x = pts[4 * t], y = pts[4 * t + 1], c = pts[4 * t + 2];
            if (ref_apply_xf(prog, P, k, &x, &y, &c, &rng[t])) { free(sx); free(dith); free(sel); free(sel_next); return -1; }
            uint32_t dst = shuffle_dest(g, wv, l, R);                /* iter.py:274-283 */
            sx[dst] = x; sy[dst] = y; sc[dst] = c;
        }
        for (int t = 0; t < nt; ++t) {
            pts[4 * t] = sx[t]; pts[4 * t + 1] = sy[t]; pts[4 * t + 2] = sc[t];
        }
        if (rd < fuse) continue;                                     /* iter.py:298-300 */
        for (int t = 0; t < nt; ++t) {
            float x = pts[4 * t], y = pts[4 * t + 1], cc = pts[4 * t + 2];
            if (has_final) {                                         /* iter.py:302-307 */
                if (ref_apply_xf(prog, P, nxf, &x, &y, &cc, &rng[t])) { free(sx); free(dith); free(sel); free(sel_next); return -1; }
            }
            float cx = fmaf(P[0], x, fmaf(P[1], y, P[2]));           /* iter.py:306-309 */
            float cy = fmaf(P[3], x, fmaf(P[4], y, P[5]));
            uint32_t ix = trunca(cx), iy = trunca(cy);               /* iter.py:313 */
            if (ix >= dim->astride || iy >= dim->ah) { ctr[1]++; continue; }   /* iter.py:315-317 */
            uint32_t gi = iy * dim->astride + ix;
            uint32_t flag = (hot[gi >> 4] >> ((gi & 15) << 1)) & 3;  /* iter.py:319-323 */
            float mult = 1.0f;
            if (flag) {                                              /* iter.py:325-329 */
                mult = hot_mult(flag);
                if (ref_mwc_next_01(&rng[t]) > 1.0f / mult) { ctr[2]++; continue; }
            }
            ctr[0]++;
            ctr[3] += plot_cell(palrow, gi, cc, dith[t], mult, atom, out4);
        }
    }
    free(sx); free(dith); free(sel); free(sel_next);
    return 0;
}

/* All slots of one launch.  The reference runs one block column per temporal sample (grid
 * (1024, n), cuburn/code/iter.py:165,184; cuburn/render.py:343-346), so every temporal sample
 * receives the same number of iterations.  The device model keeps that property for any slot
 * count: there are as many temporal samples as slots (`params` holds nslots blocks, block s
 * evaluated at ts + s*td/nslots), slot s uses block s and palette row s*64/nslots. */
int ref_iter_launch(const ref_geom *g, const ref_dim *dim, const int32_t *prog, const float *params,
                    const uint64_t *palette, ref_mwc *rng, float *points, uint32_t nslots,
                    const uint32_t *hot, uint64_t *atom, float *out4,
                    uint32_t round0, uint32_t nrounds, uint32_t fuse, uint64_t counters[4])
{
    const int nt = g->nw * g->wl;
    int pstride = prog[3];
    for (uint32_t s = 0; s < nslots; ++s) {
        uint32_t ts = s;
        if (iter_block(g, dim, prog, params + (size_t)ts * pstride, palette + (size_t)((uint64_t)s * 64 / nslots) * 256,
                       rng + (size_t)s * nt, points + (size_t)s * nt * 4, hot, atom, out4,
                       round0, nrounds, fuse, counters))
            return -1;
    }
    return 0;
}

/* cuburn/code/iter.py:420-544: drain the packed cells into the float accumulator, weighting
 * by the hot-flag multiplier that was in force while they were filled, then recompute the
 * flags from the accumulated density (monotonic map 0/1/2/3 for >128/>512/>2048;
 * SURVEY.md §8 a7 explains why the reference's swapped 1<->2 encoding is not required). */
void ref_flush(const ref_dim *dim, uint64_t *atom, float *out4, uint32_t *hot)
{
    size_t nbins = (size_t)dim->ah * dim->astride;
    for (size_t gi = 0; gi < nbins; ++gi) {
        uint32_t sh = (gi & 15) << 1;
        uint32_t flag = (hot[gi >> 4] >> sh) & 3;
        float mult = hot_mult(flag);
        uint32_t u[4];
        ref_unpack_cell(atom[gi], u);
        atom[gi] = 0;
        float *o = &out4[4 * gi];
        float m255 = mult * INV255;
        o[3] = fmaf((float)u[3], mult, o[3]);
        o[0] = fmaf((float)u[0], m255, o[0]);
        o[1] = fmaf((float)u[1], m255, o[1]);
        o[2] = fmaf((float)u[2], m255, o[2]);
        uint32_t nf = (o[3] > 128.0f) + (o[3] > 512.0f) + (o[3] > 2048.0f);
        hot[gi >> 4] = (hot[gi >> 4] & ~(3u << sh)) | (nf << sh);
    }
}

/* ------------------------------------------------------------------ flam3-style baseline */
typedef struct f3_job {
    const ref_dim *dim; const int32_t *prog; const float *params; const float *palf;
    ref_mwc rng; uint64_t nsamples; int fuse; float *hist; uint64_t accepted; int tid; int nthreads; uint32_t nts;
    struct f3_job *all; pthread_barrier_t *bar; float *out4;
    int nworkers;          /* OS threads: trajectory t is run by worker t % nworkers (jobs[0 .. nworkers) are the workers' own) */
} f3_job;

/* one trajectory (`j`: its RNG, its share of the chunks) into the histogram `hist` */
static void f3_trajectory(f3_job *j, float *hist)
{
    const int32_t *prog = j->prog;
    int pstride = prog[3], nxf = prog[1], has_final = prog[2];
    const uint32_t astride = j->dim->astride, ah = j->dim->ah;
    ref_mwc *r = &j->rng;
    float x = ref_mwc_next_11(r), y = ref_mwc_next_11(r), c = ref_mwc_next_01(r);
    int fuse = j->fuse;
    /* The job's samples are cut into nchunks = nts * k chunks of (almost) equal length, chunk c
     * belongs to temporal sample c % nts and to trajectory c % nthreads: every temporal sample gets the
     * same number of iterations (cuburn/render.py:343-346: one block column per temporal sample). */
    const uint64_t total = j->nsamples;
    uint64_t per_ts = (total + 2048ull * j->nts) / (4096ull * j->nts);
    if (per_ts < 1) per_ts = 1;
    const uint64_t nchunks = per_ts * j->nts;
    for (uint64_t chunk = (uint64_t)j->tid; chunk < nchunks; chunk += (uint64_t)j->nthreads) {
        uint32_t ts = (uint32_t)(chunk % j->nts);
        const float *P = j->params + (size_t)ts * pstride;
        const float *pal = j->palf + (size_t)((uint64_t)ts * 64 / j->nts) * 256 * 3;
        uint64_t n = total / nchunks + (chunk < total % nchunks);
        for (uint64_t i = 0; i < n + (uint64_t)fuse; ++i) {
            if (!isfinite(fabsf(x) + fabsf(y))) { x = ref_mwc_next_11(r); y = ref_mwc_next_11(r); c = ref_mwc_next_01(r); }
            int k = select_xf(prog, P, ref_mwc_next_01(r));
            ref_apply_xf(prog, P, k, &x, &y, &c, r);
            if (i < (uint64_t)fuse) continue;
            float fx = x, fy = y, fc = c;
            if (has_final) ref_apply_xf(prog, P, nxf, &fx, &fy, &fc, r);
            float cx = fmaf(P[0], fx, fmaf(P[1], fy, P[2]));
            float cy = fmaf(P[3], fx, fmaf(P[4], fy, P[5]));
            uint32_t ix = trunca(cx), iy = trunca(cy);
            if (ix >= astride || iy >= ah) continue;
            float cf = fmaf(fc, 255.0f, 0.49f * ref_mwc_next_11(r));
            int ci = (cf != cf) ? 0 : (cf >= 255.0f ? 255 : (cf <= 0.0f ? 0 : (int)rintf(cf)));
            float *o = hist + 4 * ((size_t)iy * astride + ix);
            o[0] += pal[3 * ci]; o[1] += pal[3 * ci + 1]; o[2] += pal[3 * ci + 2]; o[3] += 1.0f;
            j->accepted++;
        }
        fuse = 0;
    }
}

static void *f3_worker(void *arg)
{
    f3_job *j = arg;                               /* jobs[w], w < nworkers */
    const size_t nfl = (size_t)j->dim->ah * j->dim->astride * 4;
    j->hist = calloc(nfl, sizeof(float));          /* first touch on the worker's own NUMA node */
    /* this worker's trajectories, one after the other, into its one histogram (tests ask for up to 65536 trajectories — a
     * sample made like the GPU's, many short orbits; one OS thread and one image-sized histogram each was minutes of merging) */
    for (int t = j->tid; t < j->nthreads; t += j->nworkers) f3_trajectory(&j->all[t], j->hist);
    /* merge: every worker sums one stripe of the image over all private histograms */
    pthread_barrier_wait(j->bar);
    size_t lo = nfl * (size_t)j->tid / j->nworkers, hi = nfl * (size_t)(j->tid + 1) / j->nworkers;
    for (int t = 0; t < j->nworkers; ++t) {
        const float *h = j->all[t].hist;
        for (size_t i = lo; i < hi; ++i) j->out4[i] += h[i];
    }
    pthread_barrier_wait(j->bar);
    free(j->hist);
    return NULL;
}

/* Classic flam3-style chaos game: every walker draws its own xform each iteration
 * (no wave coherence, no point swap), one private float histogram per thread (allocated and
 * merged in parallel).  Returns the wall seconds of iterate + merge.  out4 is ADDED to. */
double ref_flam3_render(const ref_dim *dim, const int32_t *prog, const float *params, uint32_t nts,
                        const uint64_t *palette, const ref_mwc *seeds, uint32_t nseeds,
                        uint64_t nsamples, int nthreads, int fuse, float *out4, uint64_t *accepted)
{
    float *palf = malloc(sizeof(float) * 64 * 256 * 3);
    for (int i = 0; i < 64 * 256; ++i) {
        uint32_t u[4];
        ref_unpack_cell(palette[i], u);
        palf[3 * i] = (float)u[0] * INV255; palf[3 * i + 1] = (float)u[1] * INV255; palf[3 * i + 2] = (float)u[2] * INV255;
    }
    /* `nthreads` trajectories on at most 64 OS threads (up to 64 — the CPU baseline's case — one trajectory per OS thread) */
    const int nworkers = nthreads < 64 ? nthreads : 64;
    f3_job *jobs = calloc(nthreads, sizeof(f3_job));
    pthread_t *th = calloc(nworkers, sizeof(pthread_t));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, nworkers);
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = (f3_job){dim, prog, params, palf, seeds[t % nseeds], nsamples,
                           fuse, NULL, 0, t, nthreads, nts, jobs, &bar, out4, nworkers};
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nworkers; ++t) pthread_create(&th[t], NULL, f3_worker, &jobs[t]);
    uint64_t acc = 0;
    for (int t = 0; t < nworkers; ++t) pthread_join(th[t], NULL);
    for (int t = 0; t < nthreads; ++t) acc += jobs[t].accepted;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    pthread_barrier_destroy(&bar);
    free(jobs); free(th); free(palf);
    if (accepted) *accepted = acc;
    return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}
