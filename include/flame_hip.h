/*
 * flame_hip.h — C ABI of the MI355X-native fractal-flame hot path (libflame_hip.so).
 *
 * Drop-in boundary for cuburn's device path.  Each entry point cites the reference
 * interface (file:line in stevenrobertson/cuburn) it replaces.  Plain C types only:
 * no C++ / torch / HIP types cross this boundary (streams and device pointers travel
 * as void* / uint64_t).
 *
 * Status codes: 0 = ok; negative = error (FL_E_*); fl_last_error() gives text.
 * Threading: one fl_ctx per GPU per thread; calls on one ctx are not re-entrant
 * (cuburn/render.py:401-402 makes the same statement for RenderManager).
 */
#ifndef FLAME_HIP_H
#define FLAME_HIP_H

#ifndef __HIPCC_RTC__      /* hipRTC (run-time specialised iterate kernel) has no system headers; the includer typedefs */
#include <stdint.h>
#include <stddef.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define FL_ABI_VERSION 1

enum {
    FL_OK = 0,
    FL_E_INVAL = -1,      /* bad argument / malformed program                      */
    FL_E_NOMEM = -2,      /* hipErrorOutOfMemory; ctx stays usable (render.py:140-147) */
    FL_E_HIP = -3,        /* any other HIP runtime error                            */
    FL_E_NODEV = -4,      /* no usable gfx950 device                                */
    FL_E_UNSUPPORTED = -5 /* variation id / filter id not implemented               */
};

/* ------------------------------------------------------------------------------------
 * Formats that cross the boundary
 * ------------------------------------------------------------------------------------
 *
 * (1) Dimensions — cuburn/render.py:79-89 (Framebuffers.calc_dim): gutter 12,
 *     aw = w + 24, astride = ceil32(aw), ah = ceil16(h + 24).
 */
typedef struct fl_dim {
    uint32_t w, h, aw, ah, astride;
} fl_dim;

/* (2) MWC RNG state — cuburn/code/mwc.py:49-55 (mwc_st), 3 x u32, array-of-structs. */
typedef struct fl_mwc {
    uint32_t mul, state, carry;
} fl_mwc;

/* (3) Packed accumulator cell — cuburn/code/interp.py:428-429 (writer),
 *     cuburn/code/iter.py:385-389,464-468 (reader).  64-bit cell:
 *        hi[31:22] count (10 b) | hi[21:4] sum Y (18 b) | {hi[3:0],lo[31:18]} sum U (18 b)
 *        | lo[17:0] sum V (18 b)
 *     A palette entry is pre-packed as hi = (1<<22)|(y<<4), lo = (u<<18)|v, y,u,v in 0..255.
 *
 * (4) Spline rows — cuburn/code/interp.py:207-232 (GenomePacker.pack): per genome
 *     parameter one row of FL_KNOTS (=32) knot times (padded 1e9) and FL_KNOTS knot
 *     values; normalisation per cuburn/genome/use.py:129-158.
 */
#define FL_KNOTS 32
#define FL_NTEMPORAL 1024   /* cuburn/render.py:207 ntemporal_samples = the minimum number of temporal samples here: one per
                               walker slot (fl_ctx_create nslots >= 1024), or two / four per slot of 512 8-wave / 256 16-wave slots, see fl_interp */
#define FL_PAL_W 256        /* cuburn/render.py:201-202 palette surface 256 x 64 */
#define FL_PAL_H 64
#define FL_GUTTER 12        /* cuburn/render.py:77 */

/* (5) Xform program + parameter block — replaces the per-genome generated CUDA of
 *     cuburn/code/iter.py:121-149,559-575 with data interpreted by one precompiled kernel.
 *     The genome's STRUCTURE (which variations, post affines, final xform) is laid out in the
 *     parameter block itself with fixed strides, so that the kernel reaches everything about
 *     the chosen xform with ONE level of address arithmetic (no pointer chasing).
 *
 *     program header (int32 x 8):
 *       prog[0]  FL_PROG_MAGIC
 *       prog[1]  nxf        number of selectable xforms (string-sorted key order,
 *                            cuburn/genome/use.py:88-91)
 *       prog[2]  has_final  0/1 (cuburn/code/iter.py:303-307)
 *       prog[3]  pstride    floats per temporal-sample parameter block
 *       prog[4]  cdf_off    block offset of CDF[nxf] (cuburn/code/iter.py:12-30; entry
 *                            nxf-1 is stored too and is >= 1)
 *       prog[5]  xf_off     block offset of xform record 0
 *       prog[6]  xf_stride  floats per xform record (multiple of 4)
 *       prog[7]  var_stride floats per variation record (>= 2)
 *     parameter block (32-bit words; floats unless noted):
 *       [0..5]   camera xx,xy,xo,yx,yy,yo (cuburn/code/iter.py:56-79)
 *       xform record i (i = nxf is the final xform) at xf_off + i * xf_stride:
 *         [0..5]  pre affine xx,xy,xo,yx,yy,yo (cuburn/code/iter.py:81-95)
 *         [6..11] post affine (same six; unused unless the flag below says so)
 *         [12] color   [13] color_speed
 *         [14] INT: nvar | (has_post << 8)      [15] reserved
 *         variation j (sorted-name order, cuburn/code/iter.py:132) at 16 + j * var_stride:
 *           [0] INT flam3 variation number (cuburn/genome/variations.py:28-127)
 *           [1] weight   [2..] the variation's genome parameters in sorted-name order,
 *                              then its precalculated values (cuburn_amd/genome/variations.py)
 */
#define FL_PROG_MAGIC 0x464c5032 /* 'FLP2' */
#define FL_PROG_HDR 8
#define FL_MAX_XFORMS 64
#define FL_MAX_PSTRIDE 4096
#define FL_XF_HDR 16      /* words of an xform record before its first variation */

/* (6) Interpolation op list (int32 x 4 per op) — replaces the generated
 *     interp_iter_params kernel body (cuburn/code/interp.py:234-272) and the precalc
 *     snippets registered through PrecalcWrapper._code.  op = {kind, dst, a, b}:
 */
enum {
    FL_OP_SPLINE = 0,     /* dst <- catmull_rom(row a)        (interp.py:318-355)        */
    FL_OP_SPLINE_MAG = 1, /* dst <- catmull_rom_mag(row a)    (interp.py:299-316,339-353)*/
    FL_OP_CAMERA = 2,     /* dst[0..5] <- camera precalc; rows a..a+3 = rotation,
                             center.x, center.y, scale(mag)   (iter.py:56-79)            */
    FL_OP_AFFINE = 3,     /* dst[0..5] <- affine precalc; rows a..a+5 = angle, spread,
                             magnitude.x(mag), magnitude.y(mag), offset.x, offset.y
                             (iter.py:81-95)                                             */
    FL_OP_CDF = 4,        /* dst[0..b-1] <- cumulative density; rows a..a+b-1 = xform
                             weights (iter.py:12-30)                                     */
    FL_OP_RATIO2 = 5,     /* dst <- row a / (2 * row b)   julian/juliascope cn
                             (variations.py:292-294); a = dist(mag), b = power(mag)      */
    FL_OP_INVSQ = 6,      /* dst <- 1/(v*v + 1e-20), v = row a  (waves, variations.py:136-140) */
    FL_OP_PERSP = 7,      /* dst[0..2] <- mdist, sin, cos; row a = angle, row b = dist(mag)
                             (variations.py:267-273)                                     */
    FL_OP_INVSQ_MAX = 8,  /* dst <- 1/max(1e-20, v*v), v = row a (mag)  (curve, variations.py:630-634) */
    FL_OP_CONST = 9       /* dst <- the 32 bits of a (structure words: variation numbers, counts) */
};

/* ------------------------------------------------------------------------------------
 * Entry points
 * ------------------------------------------------------------------------------------ */

typedef struct fl_ctx fl_ctx;       /* RenderManager + Framebuffers state (render.py:40-170,253-262) */
typedef struct fl_genome fl_genome; /* Renderer's compiled module + packer layout (render.py:225-251) */

int fl_abi_version(void);
const char *fl_last_error(void);

/* cuburn/render.py:79-89 Framebuffers.calc_dim */
void fl_calc_dim(uint32_t w, uint32_t h, fl_dim *out);

/* cuburn/render.py:253-262 RenderManager.__init__ + :91-104 Framebuffers.__init__:
 * device, streams, walker/RNG state (persistent across frames, render.py:95-104).
 * `seeds` = nseeds x {mul,state,carry} as built by make_seeds (mwc.py:30-47);
 * nseeds must be nslots * 64 * NW + FL_PAL_H * 256 + 65536, where NW = 4, 8 or 16 is the number of
 * waves per iterate workgroup (the table's size selects it): walkers, then the palette kernel's
 * states, then the output dither's.  nslots: a multiple of 256 in [1024, 16384] — one temporal sample per
 * slot — or 512 with NW = 8 / 256 with NW = 16: every four waves of a workgroup then walk a temporal sample of
 * their own (1024 in all, 256 walkers each: the reference's geometry, cuburn/render.py:207), sharing one sort batch.  stream = a hipStream_t to run everything on (single lane), or NULL:
 * the context then owns two streams and alternates consecutive frames between them so that the
 * drain / filter / output work of frame k overlaps the iteration of frame k+1
 * (cuburn/render.py:432-433 swaps stream_a / stream_b the same way). */
int fl_ctx_create(int device, void *stream, const fl_mwc *seeds, uint32_t nseeds, uint32_t nslots, fl_ctx **out);
void fl_ctx_destroy(fl_ctx *ctx);
int fl_ctx_sync(fl_ctx *ctx);

/* cuburn/render.py:232-251 Renderer.compile/load: takes the xform program (5) and the
 * interpolation op list (6) instead of generated source. */
int fl_genome_create(fl_ctx *ctx, const int32_t *prog, uint32_t nprog,
                     const int32_t *ops, uint32_t nops, uint32_t nrows, fl_genome **out);
void fl_genome_destroy(fl_genome *g);

/* cuburn/render.py:264-285 RenderManager._copy: upload packed splines and palettes.
 * times/knots: nrows x FL_KNOTS floats; pal_rgba: npal x 256 x 4 floats;
 * pal_times: FL_KNOTS floats padded 1e9. */
int fl_genome_upload(fl_ctx *ctx, fl_genome *g, const float *times, const float *knots,
                     const float *pal_rgba, const float *pal_times, uint32_t npal);

/* cuburn/render.py:289-307 RenderManager._interp: interp_palette_flat + interp_iter_params
 * for the frame window [ts, ts+td).  The reference evaluates 1024 temporal samples and runs one
 * block column per sample (grid (1024, n), render.py:343-346), so every sample gets the same number
 * of iterations.  Here the number of temporal samples equals the number of walker slots (twice that
 * / four times that for 512 slots of 8 waves / 256 of 16): block s (s < n) is evaluated at ts + s*td/n and iterated
 * by slot s (by sub-block s % k of slot s / k), palette row r (of 64) at ts + r*td/64 is used by slots
 * [r*nslots/64, (r+1)*nslots/64) — equal weights for any nslots. */
int fl_interp(fl_ctx *ctx, fl_genome *g, uint32_t w, uint32_t h, float ts, float td);

/* Accumulation back-ends for fl_iterate. */
enum {
    FL_ACCUM_ATOMIC = 0, /* 64-bit packed global atomics, as cuburn/code/iter.py:331-411 */
    FL_ACCUM_BINNED = 1  /* LDS-staged tile binning, then packed adds per tile          */
};

/* cuburn/render.py:316-372 RenderManager._iter: clears, fuse, iteration rounds, flushes.
 * nsamples = write-enabled chaos-game iterations requested (spp*w*h, render.py:331);
 * the count actually run (rounded up to whole launches) is returned in *nsamples_run.
 * fuse = write-disabled iterations per walker at frame start (reference: 256). */
int fl_iterate(fl_ctx *ctx, fl_genome *g, uint32_t w, uint32_t h, double nsamples,
               uint32_t fuse, int accum_mode, uint64_t *nsamples_run);

/* cuburn/filters.py:45-53,56-95,98-108,138-174 Filter.apply — one call per filter, result
 * left in the front buffer (filters.py:29-35).  params are the host-derived scalars. */
enum {
    FL_FILT_YUV = 0,       /* yuv_to_rgb       code/filters.py:71-77   params: none                 */
    FL_FILT_BILATERAL = 1, /* DE chain         filters.py:62-95        sstd,cstd,dstd,dpow,gspeed   */
    FL_FILT_LOGSCALE = 2,  /* logscale         code/filters.py:41-53   k1,k2                        */
    FL_FILT_COLORCLIP = 3, /* colorclip        code/filters.py:354-412 vib,highpow,gam,lin,lingam   */
    FL_FILT_SMEARCLIP = 4, /* smearclip chain  filters.py:142-163      width,gam_m_1,lin,lingam     */
    FL_FILT_HALOCLIP = 5,  /* haloclip chain   filters.py:113-130      gam_m_1                      */
    FL_FILT_PLAINCLIP = 6, /* plainclip        code/filters.py:332-350 gam_m_1,lin,lingam,brightness*/
    FL_FILT_LOGENCODE = 7  /* logencode        code/filters.py:81-90   degamma                      */
};
int fl_filter(fl_ctx *ctx, int filter_id, uint32_t w, uint32_t h, const float *params, uint32_t nparams);

/* cuburn/output.py:83-88,120-125,150-158,212-219,323-338 (convert + copy of every Output class):
 * f32 -> pixel format with dither (cuburn/code/output.py:7-236), gutter cropped; async D2H into
 * host_out (or device copy when dev_out != 0).  fl_output_bytes gives the frame's size. */
enum {
    FL_OUT_RGBA8 = 0,     /* f32_to_rgba_u8    code/output.py:20-44    u8  [h][w][4]                        */
    FL_OUT_RGBA16 = 1,    /* f32_to_rgba_u16   code/output.py:47-71    u16 [h][w][4]                        */
    FL_OUT_YUV444P = 2,   /* f32_to_yuv444p    code/output.py:75-102   u8  [3][h][w], JPEG full range       */
    FL_OUT_YUV444P10 = 3, /* f32_to_yuv444p10  code/output.py:106-134  u16 [3][h][w], peak 1023             */
    FL_OUT_YUV420P10 = 4, /* f32_to_yuv420p10  code/output.py:138-190  u16 Y[h][w] Cb[h/2][w/2] Cr[h/2][w/2]; w, h even */
    FL_OUT_YUV444P12 = 5  /* f32_to_yuv444p12  code/output.py:194-221  u16 [3][h][w], Rec.709 studio swing  */
};
size_t fl_output_bytes(uint32_t w, uint32_t h, int fmt);   /* 0 for an unknown format */
int fl_output(fl_ctx *ctx, uint32_t w, uint32_t h, int fmt, void *host_out, uint64_t dev_out);

/* cuburn/code/sort.py:443-504 Sorter.sort: one radix pass over n 32-bit keys on the device —
 * dst = src ordered by the `nbits` (1..10) bits from lo_bit up, keys with equal digits in their
 * original order (stable: passes from the low digit up compose into a full sort; the reference's
 * pass is not stable and its multi-pass sort is marked broken, sort.py:437-441,455-458).
 * ignore_max: keys equal to 0xffffffff are dropped (sort.py:449-452); *nvalid (optional, makes the
 * call synchronous) receives the number of keys written.  dst and src are device addresses and must
 * not overlap; every pass runs on the context's FIRST stream (the caller's, if one was given to
 * fl_ctx_create), so consecutive passes are ordered.  Not used by the render path — like
 * the reference's sorter (imported by render.py:19, never called). */
int fl_sort_u32(fl_ctx *ctx, uint64_t dst_dev, uint64_t src_dev, uint32_t n, uint32_t lo_bit, uint32_t nbits, int ignore_max,
                uint32_t *nvalid);

/* cuburn/render.py:404,430 timing_event / DurationEvent: fl_frame_begin opens a frame and returns
 * its id; fl_output closes it.  fl_frame_ms blocks until that frame is done and gives the ms
 * between the two (DurationEvent.time, render.py:26-38); fl_frame_query is DurationEvent.query
 * (1 done, 0 running).  Up to 4 frames may be in flight. */
int fl_frame_begin(fl_ctx *ctx, uint32_t *frame_id);
int fl_frame_ms(fl_ctx *ctx, uint32_t frame_id, float *ms);
int fl_frame_query(fl_ctx *ctx, uint32_t frame_id);

/* cuburn/render.py:93 PageLockedMemoryPool: pinned host memory for h_out so that the D2H copy of
 * fl_output is truly asynchronous. */
void *fl_host_alloc(size_t nbytes);
void fl_host_free(void *p);

/* ---- measurement taps (bench.py / tests only) ---- */
/* HIP-event times accumulated since the last fl_timings_reset: iterate kernels; drain kernels
 * (tile accumulate + flush); filter kernels; number of iterate launches.  Both calls sync. */
int fl_timings_reset(fl_ctx *ctx);
int fl_timings(fl_ctx *ctx, float *iter_ms, float *flush_ms, float *filter_ms, uint32_t *niter_launches);
/* The same, split further: ms[0] iterate kernels, [1] tile accumulate, [2] flush, [3] all fl_filter calls,
 * [4] the DE proper — the eight direction kernels, the first normalising the accumulator, the last un-normalising it with a
 * following logscale / colorclip riding along — recorded around the launches themselves, whichever call flushes them
 * (the colorclip that follows, another filter, fl_output, a debug tap); [5] unused since round 5 (always 0). */
int fl_timings_detail(fl_ctx *ctx, float ms[6]);
/* Which iterate kernel actually ran since the last fl_timings_reset: out[0] launches of the kernel compiled for the
 * genome's structure (hipRTC; the counterpart of the module the reference compiles per genome, cuburn/render.py:232-236),
 * out[1] launches of the precompiled interpreter kernel (the fallback); out[2] walker slots, out[3] waves per slot. */
int fl_launch_stats(fl_ctx *ctx, uint32_t out[4]);
/* The device's streaming ceiling as this library can reach it: a float4 copy of `nbytes` (read + write; at least 1 GiB each
 * way, so that the 256 MiB Infinity Cache does not serve it) with non-temporal loads and stores, `iters` launches between two HIP
 * events; *ms = milliseconds per copy.  The denominator of bench.py's "DE >= 60 % of measured HBM bandwidth" (BASELINE.json,
 * SURVEY.md 8d); the reference has no such measurement. */
int fl_measure_copy(int device, size_t nbytes, int iters, float *ms);

/* ---- debug taps (tests only): read/write device state ---- */
enum {
    FL_BUF_FRONT = 0,   /* float4[nbins]  accumulator / filter result (render.py:44-48)  */
    FL_BUF_BACK = 1,    /* float4[nbins]                                                   */
    FL_BUF_PARAMS = 2,  /* float[ntemporal * pstride] interpolated parameter blocks (one per temporal sample = per slot; two / four per slot of 512 8-wave / 256 16-wave slots) */
    FL_BUF_PALETTE = 3, /* u64[FL_PAL_H * FL_PAL_W] packed palette (interp.py:409-433)     */
    FL_BUF_POINTS = 4,  /* float4[nwalkers] walker points (render.py:102-104)              */
    FL_BUF_SEEDS = 5,   /* fl_mwc[nwalkers]                                                */
    FL_BUF_ATOM = 6,    /* u64[nbins] packed integer accumulator of the current side      */
    FL_BUF_HOT = 7,     /* u32[nbins/16] hot-pixel flags of the current side               */
    FL_BUF_SIDE = 8     /* float4[nbins] side buffer                                       */
};
int fl_read_buffer(fl_ctx *ctx, fl_genome *g, int which, void *host_dst, size_t nbytes);
int fl_write_buffer(fl_ctx *ctx, fl_genome *g, int which, const void *host_src, size_t nbytes);

/* Device address and size of one of the buffers above (current frame's lane), after waiting for
 * all queued work of the context.  This is the hook for the one exchange step of sample-sharded
 * rendering of a single frame: every rank iterates its share of the samples,
 * the caller sums FL_BUF_FRONT across ranks with its own collective (RCCL all-reduce on its own
 * stream), synchronises that stream, and continues with fl_filter / fl_output.  The reference has
 * no such step (distribute.py:149-163 shards whole frames only). */
int fl_buffer_ptr(fl_ctx *ctx, fl_genome *g, int which, void **dev_ptr, size_t *nbytes);
/* The same without blocking the host (round 5): the address is handed out at once, and what the caller queues on a stream of its
 * own is ordered against the context's work with fl_stream_dependency.  The reference orders its two streams the same way —
 * an event recorded on one, stream.wait_for_event on the other, no host wait inside a frame (cuburn/render.py:358-364,419-430;
 * distribute.py:107-122 is its double-buffered loop). */
int fl_buffer_ptr_async(fl_ctx *ctx, fl_genome *g, int which, void **dev_ptr, size_t *nbytes);
/* Order the context's current lane and a caller's HIP stream (`stream`: a hipStream_t, e.g. torch's current stream, on which the
 * caller runs a collective over buffers obtained above) without a host wait.  ctx_waits = 0: everything queued on the lane so far
 * (deferred filter steps included) happens before whatever the caller queues on `stream` next; ctx_waits = 1: everything queued
 * on `stream` so far happens before whatever the context queues on the lane next. */
int fl_stream_dependency(fl_ctx *ctx, void *stream, int ctx_waits);
/* Make the current lane's frame buffers large enough for a w x h image now (they only ever grow; what they held is lost when
 * they do, so: before a frame's fl_iterate).  The sample-sharded path reserves the height that makes the accumulator's rows a
 * multiple of the rank count: its reduce-scatter then runs on the accumulator where it lies (the rows behind the frame's own are
 * summed and never looked at) instead of on a zero-padded copy per frame.  The reference's accumulator is allocated once, for the
 * largest frame (cuburn/render.py:44-48, :107-113). */
int fl_reserve(fl_ctx *ctx, uint32_t w, uint32_t h);

/* Single-launch taps for bit-exact tests: run `nrounds` rounds (first `fuse` write-disabled)
 * for every slot with the given global round counter, no clears, no flush. */
int fl_debug_iter_launch(fl_ctx *ctx, fl_genome *g, uint32_t w, uint32_t h, uint32_t round0,
                         uint32_t nrounds, uint32_t fuse, int accum_mode);
int fl_debug_flush(fl_ctx *ctx, uint32_t w, uint32_t h);
int fl_debug_clear(fl_ctx *ctx, uint32_t w, uint32_t h, int reset_points);
int fl_debug_clear_hot(fl_ctx *ctx, uint32_t w, uint32_t h);
/* Point-shuffle tap: out[dst_thread] = src_thread after one swap with round counter `round`. */
int fl_debug_shuffle(fl_ctx *ctx, uint32_t round, uint32_t *out256);
/* Xform tap: apply xform `xfi` (nxf = the final xform) of temporal sample `ts` once to n points
 * {x, y, color, unused} with one RNG state each; results overwrite the inputs. */
int fl_debug_apply_xf(fl_ctx *ctx, fl_genome *g, uint32_t ts, int xfi, uint32_t n, float *xyzw, fl_mwc *rng);
/* Compile (only) the iterate kernel specialised for a genome structure, as fl_iterate does on a genome's
 * first launch — the counterpart of cuburn/render.py:232-236 Renderer.compile.  Needs libhiprtc but no
 * GPU; FL_E_UNSUPPORTED if hipRTC is not installed.  nw = 4 | 8 | 16, acc = 0 (atomic), 1 / 3 (binned narrow / wide);
 * count: bit 0 = the sample counters, bit 1 = a temporal sample per four waves (nw = 8 / 16: two / four per workgroup). */
int fl_rtc_compile_check(const int32_t *prog, uint32_t nprog, const int32_t *ops, uint32_t nops, int nw, int count, int acc,
                         char *log, size_t log_bytes);
/* Counters of the last iterate: accepted (written) samples, out-of-frame, roulette-dropped, spills. */
int fl_debug_counters(fl_ctx *ctx, uint64_t out4[4]);

/* ---- run-time switches (environment; read when a context is created / a kernel is first launched) ----
 * Every switch is an A/B lever of a documented measurement, none changes results beyond float summation order:
 *   FLAME_LANES=n            stream lanes, 1..4 (default 2: consecutive frames alternate; 1 for kernel timing — profiles/ are taken with it;
 *                            3 and 4 measure the same as 2, profiles/r05_lanes.txt)
 *   FLAME_NO_INTRA_OVERLAP=1 the launches of a multi-launch frame strictly in series on one stream
 *   FLAME_RTC=0              always the precompiled interpreter iterate kernel (no hipRTC per-genome kernel)
 *   FLAME_RTC_FLAGS=...      extra options for the hipRTC compile; FLAME_RTC_DUMP=<dir> keeps its source / assembly
 *   FLAME_BIN_ROUNDS=n       rounds per sorted batch of the sample log (default 16)
 *   FLAME_BIN_PARTS=n        workgroups per tile of the tile accumulate (default: by image size)
 *   FLAME_BIN_GANG=n         adjacent tiles per XCD gang of the tile accumulate (default 32 above 512 tiles, else 0 = off)
 *   FLAME_BIN_WIDE=1         256x64 accumulate tiles for every image size
 *   FLAME_LAUNCH_ROUNDS=n    most write-enabled rounds per binned iterate launch, 16..4096 (default: the reference's growing batches
 *                            capped at 1024 rounds, or at 2304 when that saves the frame a launch)
 *   FLAME_DE_ORDER=d|dddddddd tile order of the DE kernels, one digit for all or one per direction (0 per-XCD column-major runs,
 *                            1 row-major, 2 row-major in runs per XCD; default: by direction and image size, de.hip)
 * Compile-time timing builds (-DDE_X_*, -DACC_X_*) produce wrong pictures and exist only in libraries built for tools/. */

#ifdef __cplusplus
}
#endif
#endif /* FLAME_HIP_H */
