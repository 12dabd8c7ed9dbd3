/*
 * flame_ref.h — CPU restatement ("oracle") of cuburn's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cuburn_amd/ may include, link, import or
 * execute anything in oracle/; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * Parity status: the reference has NO CPU renderer and no test of iter / flush / any
 * filter => for those functions the oracle is "parity unpinned"
 * by reference tests.  What IS pinned against reference-generated golden vectors
 * (tests/golden/, made by tests/golden/make_golden.py importing the reference's host
 * code): MWC streams and seed layout, calc_dim, spline normalisation + Catmull-Rom
 * evaluation, packer knot arrays, the point-shuffle permutation (helpers/shuf.py),
 * host-side filter scalars, profile frame times.
 *
 * All file:line citations are into /root/reference (stevenrobertson/cuburn).
 */
#ifndef FLAME_REF_H
#define FLAME_REF_H
#include <stdint.h>
#include <stddef.h>

typedef struct { uint32_t mul, state, carry; } ref_mwc;
typedef struct { uint32_t w, h, aw, ah, astride; } ref_dim;

/* Block geometry of the device model.  The reference runs 8 warps x 32 lanes with
 * the swap of cuburn/code/iter.py:274-294 (ref_shuffle = 1).  The MI355X kernel runs
 * nw waves x 64 lanes with the generalised three-phase swap (ref_shuffle = 0). */
typedef struct { int nw, wl, ref_shuffle; } ref_geom;

/* cuburn/code/mwc.py:56-77 */
uint32_t ref_mwc_next(ref_mwc *s);
float ref_mwc_next_01(ref_mwc *s);
float ref_mwc_next_11(ref_mwc *s);
void ref_mwc_stream(ref_mwc *s, uint32_t n, uint32_t *out);
void ref_mwc_sums(ref_mwc *s, uint32_t nthreads, uint32_t rounds, uint64_t *sums);

/* cuburn/render.py:79-89 */
void ref_calc_dim(uint32_t w, uint32_t h, ref_dim *out);

/* cuburn/code/iter.py:274-278 / helpers/shuf.py:75-80 */
void ref_shuffle_perm(const ref_geom *g, uint32_t round, uint32_t *out);

/* cuburn/code/interp.py:284-367 + util.py:219-230 */
float ref_catmull_rom(const float *times, const float *knots, float t, int mag);

/* cuburn/code/interp.py:372-433 */
void ref_interp_palette(const float *pal_rgba, const float *pal_times, uint32_t npal,
                        float ts, float td, ref_mwc *rng64x256, uint64_t *out);

/* cuburn/code/iter.py:121-149 + variations.py */
int ref_apply_xf(const int32_t *prog, const float *P, int xfi, float *x, float *y, float *color, ref_mwc *r);
/* ... for n points {x, y, color, unused} with one RNG state each, results in place (one call per variation test instead of one per point) */
int ref_apply_xf_n(const int32_t *prog, const float *P, int xfi, uint32_t n, float *xyzw, ref_mwc *r);
int ref_var_supported(int id);

/* cuburn/code/iter.py:157-418 (device model, deterministic) */
int ref_iter_launch(const ref_geom *g, const ref_dim *dim, const int32_t *prog, const float *params,
                    const uint64_t *palette, ref_mwc *rng, float *points, uint32_t nslots,
                    const uint32_t *hot, uint64_t *atom, float *out4,
                    uint32_t round0, uint32_t nrounds, uint32_t fuse, uint64_t counters[4]);

/* cuburn/code/iter.py:420-544 */
void ref_flush(const ref_dim *dim, uint64_t *atom, float *out4, uint32_t *hot);
void ref_plot_samples(uint32_t n, const uint32_t *gi, const float *cc, const float *dither, const uint32_t *row,
                      const float *mult, const uint64_t *palette, uint64_t *atom, float *out4);
void ref_unpack_cell(uint64_t cell, uint32_t out[4]);

/* flam3-style per-sample-selection chaos game: CPU baseline */
double ref_flam3_render(const ref_dim *dim, const int32_t *prog, const float *params, uint32_t nts,
                        const uint64_t *palette, const ref_mwc *seeds, uint32_t nseeds,
                        uint64_t nsamples, int nthreads, int fuse, float *out4, uint64_t *accepted);

/* cuburn/code/filters.py (all), cuburn/code/color.py:25-40 */
void ref_yuv_to_rgb(const ref_dim *d, float *dst, const float *src);
void ref_den_blur(const ref_dim *d, float *dst, const float *src4, int pattern, int upsample, const float *coefs);
void ref_den_blur_1c(const ref_dim *d, float *dst, const float *src1, int pattern, int upsample, const float *coefs);
void ref_full_blur(const ref_dim *d, float *dst, const float *src4, int pattern, int upsample, const float *coefs);
void ref_bilateral(const ref_dim *d, float *dst, const float *src4, const float *blur1, int pattern, int radius,
                   float sstd, float cstd, float dstd, float dpow, float gspeed);
void ref_bilateral_chain(const ref_dim *d, float *front, float *back, float *side, const float *coefs,
                         float sstd, float cstd, float dstd, float dpow, float gspeed);
void ref_logscale(const ref_dim *d, float *buf, float k1, float k2);
void ref_colorclip(const ref_dim *d, float *buf, float vib, float highpow, float gam, float lin, float lingam);
void ref_smearclip_chain(const ref_dim *d, float *front, float *back, float *side, const float *coefs,
                         float gam_m_1, float lin, float lingam);
void ref_haloclip_chain(const ref_dim *d, float *front, float *back, float *side, const float *coefs, float gam_m_1);
void ref_plainclip(const ref_dim *d, float *buf, float gam_m_1, float lin, float lingam, float brightness);
void ref_logencode(const ref_dim *d, float *dst, const float *src, float degamma);

/* cuburn/code/output.py:7-236; fmt 0 rgba8, 1 rgba16, 2 yuv444p, 3 yuv444p10, 4 yuv420p10, 5 yuv444p12 */
void ref_f32_to_rgba(const ref_dim *d, const float *src, ref_mwc *rng, uint32_t nrng, int fmt, void *dst);

#endif
