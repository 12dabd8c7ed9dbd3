// kernels.h — host-callable launchers of the gfx950 kernels (internal to libflame_hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/flame_hip.h"

typedef unsigned long long u64;

// hipFuncSetAttribute is per DEVICE: raise a kernel's dynamic-LDS cap once on every device it is
// launched on (a process may hold contexts on several GPUs).  `done` is a per-kernel bit mask.
static inline void ensure_max_dynamic_lds(const void *fn, unsigned long long &done)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;
    if (done >> dev & 1ull) return;
    hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (dev != 63) done |= 1ull << dev;
}

// iter.hip
void launch_iter(hipStream_t st, int nw, bool count, int acc, uint32_t nslots,
                 const int32_t *prog, const float *params, const u64 *palette, fl_mwc *rng,
                 float4 *points, const uint32_t *hot, u64 *atom, float *out4, u64 *counters,
                 uint32_t astride, uint32_t aheight, uint32_t round0, uint32_t nrounds, uint32_t fuse,
                 uint32_t tiles_x, uint32_t nbins, uint32_t rounds_per_batch, uint32_t nbatch_total,
                 uint32_t *log, uint32_t *dir,
                 hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, uint32_t sub_log2 = 0);      // sub_log2: 8- / 16-wave workgroups of 2 / 4 temporal samples (iter_body)
void launch_flush(hipStream_t st, u64 *atom, float4 *out, uint32_t *hot, uint32_t nbins, bool use_hot);
void launch_clear_frame(hipStream_t st, float4 *front, u64 *atom, uint32_t *hot, u64 *counters, float4 *points,
                        uint32_t nbins, uint32_t npoints);      // npoints = 0: the walkers are left alone
void launch_shuffle_tap(hipStream_t st, int nw, uint32_t *out, uint32_t round);
void launch_apply_xf_tap(hipStream_t st, const int32_t *prog, const float *params, uint32_t ts, int xfi,
                         uint32_t n, float4 *pts, fl_mwc *rng);

// rtc.hip: structure of a genome as the run-time specialised iterate kernel needs it (include/flame_hip.h (5))
#include <string>
#include <vector>
struct IterSpec {
    int nxf = 0, has_final = 0, pstride = 0, cdf_off = 0, xf_off = 0, xf_stride = 0, var_stride = 0;
    std::vector<int> nvar, post;                 // per record (selectable xforms, then the final xform)
    std::vector<std::vector<int>> vids;          // flam3 variation numbers in application order
};
bool rtc_available();
unsigned rtc_epoch();      // changes when cached modules were unloaded: function handles obtained before are void
int rtc_compile(const IterSpec &spec, int nw, bool count, int acc, std::vector<char> *code, std::string *err, const char *extra_opt = nullptr, uint32_t sub_log2 = 0);
int rtc_iter_kernel(int device, const IterSpec &spec, int nw, uint32_t nslots, bool count, int acc, hipFunction_t *fn, std::string *err, uint32_t sub_log2 = 0);
// launch_iter through a run-time compiled kernel (same arguments)
void launch_iter_fn(hipStream_t st, hipFunction_t fn, int nw, int acc, uint32_t nslots,
                    const int32_t *prog, const float *params, const u64 *palette, fl_mwc *rng,
                    float4 *points, const uint32_t *hot, u64 *atom, float *out4, u64 *counters,
                    uint32_t astride, uint32_t aheight, uint32_t round0, uint32_t nrounds, uint32_t fuse,
                    uint32_t tiles_x, uint32_t nbins, uint32_t rounds_per_batch, uint32_t nbatch_total,
                    uint32_t *log, uint32_t *dir, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t sub_log2 = 0);

// binned.hip
void launch_accum_tiles(hipStream_t st, const uint32_t *log, const uint32_t *dir, const u64 *palette,
                        u64 *atom, float *out4, uint32_t tiles_x, uint32_t nbins, uint32_t nparts,
                        uint32_t nbatch_total, uint32_t batch_records, uint32_t nslots,
                        uint32_t astride, uint32_t aheight, bool wide);

// interp.hip
void launch_interp_palette(hipStream_t st, fl_mwc *rng_pal, const float *ptimes, const float4 *pals,
                           float ts, float tstep, u64 *out);
void launch_interp_params(hipStream_t st, float *params, const float *times, const float *knots,
                          const int32_t *ops, uint32_t nops, uint32_t pstride, uint32_t nts, float ts, float tstep,
                          fl_dim dim, bool zero_first);

// filters.hip
void launch_yuv_to_rgb(hipStream_t st, fl_dim d, float4 *dst, const float4 *src);
void launch_den_blur_1c(hipStream_t st, fl_dim d, float *dst, const float *src, int pattern, int upsample, const float *coefs7);
void launch_full_blur(hipStream_t st, fl_dim d, float4 *dst, const float4 *src, int pattern, int upsample, const float *coefs7);
void launch_logscale(hipStream_t st, fl_dim d, float4 *buf, float k1, float k2);
void launch_colorclip(hipStream_t st, fl_dim d, float4 *buf, float vib, float highpow, float gam, float lin, float lingam);
void launch_gamma_full_hi(hipStream_t st, fl_dim d, float4 *dst, const float4 *src);
void launch_smearclip(hipStream_t st, fl_dim d, float4 *buf, const float4 *smear, float gam_m_1, float lin, float lingam);
void launch_apply_gamma(hipStream_t st, fl_dim d, float *dst, const float4 *src, float gamma);
void launch_haloclip(hipStream_t st, fl_dim d, float4 *buf, const float *den, float gam_m_1);
void launch_plainclip(hipStream_t st, fl_dim d, float4 *buf, float gam_m_1, float lin, float lingam, float brightness);
void launch_logencode(hipStream_t st, fl_dim d, float4 *dst, const float4 *src, float degamma);

// de.hip
// tone filters that ride along with the last DE direction (filters.py default chains: DE -> logscale -> colorclip)
struct DeTail { int do_log; float k1, k2; int do_clip; float vib, highpow, gam, lin, lingam; int order; };
void launch_de_dir(hipStream_t st, fl_dim d, int pattern, float4 *Nout, const float4 *N, const float *coefs7,
                   float sstd, float cstd, float dstd, float dpow, float gspeed, int in_mode = 1, const DeTail *tail = nullptr);
// (pattern 0 normalises the raw accumulator as it stages it — in_mode 1, or 2: after yuv -> rgb —; pattern 7 un-normalises and
// applies `tail`'s tone filters as it stores; round 5 removed the separate normalise / finish kernels, the persistent and the
// overlapped eight-direction launches of round 4 (de_chain.hip: git history, commit 58f1875) and round 1's blur + packed-math pair)

// output.hip
void launch_f32_to_rgba(hipStream_t st, fl_dim d, const float4 *src, fl_mwc *rng, uint32_t nrng, int fmt, void *dst);

int launch_measure_copy(size_t nbytes, int iters, float *ms);      // the device's streaming copy rate (fl_measure_copy)

// sort.hip: one stable radix pass (nbits <= 10 at lo_bit) over n keys; hist = scratch of
// sort_scratch_words() words, chunk_tot = *chunk_words words (the last one receives the number of keys kept)
int launch_sort_pass(hipStream_t st, uint32_t *dst, const uint32_t *src, uint32_t n, uint32_t lo_bit, uint32_t nbits,
                     int ignore_max, uint32_t *hist, uint32_t *chunk_tot, uint32_t *total_dev);
size_t sort_scratch_words(uint32_t n, uint32_t nbits, size_t *chunk_words);
