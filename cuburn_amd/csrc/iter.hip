// iter.hip — chaos-game iteration, packed-cell flush and hot-pixel flags for gfx950.
//
// Replaces the run-time generated `iter` / `flush_atom` kernels of the reference
// (cuburn/code/iter.py:157-418, :420-544) with precompiled kernels that interpret the
// xform program of include/flame_hip.h (5).  MI355X mapping:
//   * workgroup = NW waves x 64 lanes, one walker per lane, bound to a persistent slot
//     (slot = blockIdx.x).  The reference runs one block column per temporal sample (grid
//     (1024, n), iter.py:165,184; render.py:343-346), i.e. every temporal sample gets the same
//     number of iterations; here there are as many temporal samples as slots (fl_interp
//     evaluates nslots parameter blocks at ts + s*td/nslots), slot s uses block s and palette
//     row s*64/nslots — equal weights for any slot count;
//   * xform selection is per WAVE: every lane draws, lane 0's draw is broadcast with
//     v_readfirstlane, so the xform branch and all parameter loads are scalar;
//   * walkers change wave every round through a double-buffered LDS swap (one barrier per
//     round), three-phase destination pattern so no pair shares a wave 3 rounds running;
//   * samples are accumulated as 64-bit packed cells with global atomics whose returned
//     value is examined one round later (latency hidden) to drain nearly-full cells.
//
// This file is compiled twice: ahead of time into libflame_hip.so (one kernel that INTERPRETS the
// genome's structure), and at run time by hipRTC (FL_RTC, rtc.hip) with the structure of ONE genome
// as compile-time tables — the native counterpart of the reference compiling a CUDA module per
// genome (cuburn/render.py:232-236, cuburn/code/iter.py:559-575): the variation dispatch, the
// variation loop, the post-affine test and the final-xform test disappear from the round loop.
// Both builds execute the same arithmetic in the same order (bit-identical results).
#include "variations.h"
#ifndef FL_RTC
#include "kernels.h"
#include <hip/hip_ext.h>
#endif

#ifndef FL_SCAN_SERIAL_MAX
#define FL_SCAN_SERIAL_MAX 8u
#endif
#ifndef FL_SPLIT_MAX_XF
#define FL_SPLIT_MAX_XF 9         /* per-genome kernels with more xforms keep a single copy of the walk */
#endif
#ifndef FL_CNT_SETS
#define FL_CNT_SETS 3          /* sets of tile counters per 4-wave workgroup (LDS: an array of tiles + 1 words each) */
#endif
#ifndef FL_ITER_PRIO
#define FL_ITER_PRIO 2         /* wave priority of the walk part of a round (0: no priority changes) */
#endif
#ifndef FL_LOG_NT
#define FL_LOG_NT 1          /* the sample log leaves with non-temporal stores: written once, read by k_accum_tiles much later (k_accum_tiles 380 -> 347 us, k_iter unchanged) */
#endif
#ifndef FL_ITER_MERGE_MAX_XF
#define FL_ITER_MERGE_MAX_XF 4 /* per-genome binned kernels of at most this many xforms run a round's plot inside the NEXT round's xform block
                                  (same results; 0: never.  cfg2, 3 xforms: k_iter_spec 653 -> 627 us; 8 xforms + final: -0.3 ... -1.2 %) */
#endif
#ifndef FL_SORT_FULL_COPY
#define FL_SORT_FULL_COPY 1
#endif
#ifndef FL_SORT_LOCAL_TID
#define FL_SORT_LOCAL_TID 1
#endif
#ifndef FL_SCATTER_DEPTH
#define FL_SCATTER_DEPTH 4     /* returning cursor adds a thread keeps in flight in the scatter */
#endif
#ifndef FL_ITER_ROT3
#define FL_ITER_ROT3 1
#endif
#ifndef FL_CNT_SETS_BIG
#define FL_CNT_SETS_BIG 2      /* the same for 8- and 16-wave workgroups (many tiles: scanning more sets costs more than it saves) */
#endif

template <int NW>
__device__ __forceinline__ uint32_t shuffle_dest(uint32_t w, uint32_t l, uint32_t phase) {
    uint32_t sh = l + (phase == 1 ? l / NW : 0u) + (phase == 2 ? l / (NW * NW) : 0u);
    return ((w + sh) % NW) * 64u + l;
}

// The first 18 words of an xform record (include/flame_hip.h (5)): affines, colour, structure
// word, and the number + weight of the first variation.  Held in SGPRs.
struct XfHead { float f[16]; int vid0; float w0; };
#ifndef FL_XTAB_BYTES
#define FL_XTAB_BYTES 256      /* LDS behind everything else: the per-xform operand table (kTab), 16 xforms */
#endif

__device__ __forceinline__ XfHead load_head(const float *__restrict__ xf) {
    XfHead h;
#pragma unroll
    for (int i = 0; i < 16; ++i) h.f[i] = xf[i];
    h.vid0 = __float_as_int(xf[FL_XF_HDR]);
    h.w0 = xf[FL_XF_HDR + 1];
    return h;
}

// cuburn/code/iter.py:121-149: pre affine, sum of variations, optional post affine, colour
// blend.  The record is wave-uniform; `h` was loaded a round ahead (its choice depends only on
// the RNG), so the s_load latency of the record is off the critical path; only parameters of
// parametric variations and variations beyond the first are loaded on demand.
__device__ __forceinline__ void apply_xf(const XfHead &h, const float *__restrict__ xf, int var_stride,
                                         float &x, float &y, float &c, mwc_t &r)
{
    const int word14 = __float_as_int(h.f[14]);
    const int nvar = word14 & 0xff;
    float tx = fmaf(h.f[0], x, fmaf(h.f[1], y, h.f[2]));
    float ty = fmaf(h.f[3], x, fmaf(h.f[4], y, h.f[5]));
    float ox = -0.0f, oy = -0.0f;          // -0 + v = v for every v (IEEE): the first variation's add folds away (the CPU model of the tests starts at -0 as well)
    if (nvar > 0) apply_variation(h.vid0, h.w0, xf + FL_XF_HDR + 2, xf, tx, ty, ox, oy, r);
    for (int j = 1; j < nvar; ++j) {
        const float *__restrict__ v = xf + FL_XF_HDR + j * var_stride;
        apply_variation(__float_as_int(v[0]), v[1], v + 2, xf, tx, ty, ox, oy, r);
    }
    if (word14 & 0x100) {
        const float qx = fmaf(h.f[6], ox, fmaf(h.f[7], oy, h.f[8]));
        const float qy = fmaf(h.f[9], ox, fmaf(h.f[10], oy, h.f[11]));
        ox = qx; oy = qy;
    }
    const float csp = h.f[13];
    c = fmaf(c, 1.0f - csp, h.f[12] * csp);
    x = ox; y = oy;
}

#ifdef FL_RTC
// Structure tables of the genome this translation unit is compiled for (generated header
// "flame_spec.h": rtc.hip): FL_SPEC_NXF selectable xforms + FL_SPEC_FINAL, per record its number of
// variations, whether it has a post affine, and its variation numbers in order.
#include "flame_spec.h"

// FL_EARLY_TAIL (round 5): kernels whose records are fetched per round ask for the ten words behind the record's head — the first
// variation's parameters and the second one's number and weight — a round ahead, together with the head, instead of at the top of the
// xform's own block ~10 instructions before their use: a scalar load there was a scalar-cache latency per round that four waves per
// SIMD only partly cover, and every lgkmcnt wait of the block waited for it.  cfg4 (eight xforms of two variations): k_iter_spec
// 1.089 -> 1.019 ms, 30 -> 13 scalar loads and 40 -> 30 lgkmcnt waits in cfg5's loop (profiles/r05_early_tail.txt).  Same values from
// the same words: bit-identical (the per-genome kernel is held to the interpreter in tests/).  -DFL_EARLY_TAIL=0: on demand, as before.
#ifndef FL_EARLY_TAIL
#define FL_EARLY_TAIL 1
#endif
constexpr int kTailFirst = FL_XF_HDR + 2, kTailWords = 10;
struct XfTail { float w[kTailWords]; };
// a variation's parameters: word `base + i` of the record, from the registers where the tail holds it
struct VTail {
    const float *__restrict__ p; const XfTail *t; int base;
    __device__ __forceinline__ float operator[](int i) const {
        const int wd = base + i;
        return (FL_EARLY_TAIL && t && wd >= kTailFirst && wd < kTailFirst + kTailWords) ? t->w[wd - kTailFirst] : p[wd];
    }
};
template <int I, int J>
__device__ __forceinline__ void spec_variations(const float *__restrict__ xf, float w0, float &tx, float &ty,
                                                float &ox, float &oy, mwc_t &r, const XfTail *tl = nullptr)
{
    if constexpr (J < kSpecNvar[I]) {
        constexpr int B = FL_XF_HDR + J * FL_SPEC_VAR_STRIDE;
        const VTail v = {xf, tl, B + 2}, vh = {xf, tl, B};
        apply_variation_body(kSpecVid[I][J], J == 0 ? w0 : vh[1], v, xf, tx, ty, ox, oy, r);
        spec_variations<I, J + 1>(xf, w0, tx, ty, ox, oy, r, tl);
    }
}

// apply_xf with the structure of record I known at compile time (same arithmetic, same order)
template <int I>
__device__ __forceinline__ void spec_apply_xf(const XfHead &h, const float *__restrict__ xf,
                                              float &x, float &y, float &c, mwc_t &r)
{
    float tx = fmaf(h.f[0], x, fmaf(h.f[1], y, h.f[2]));
    float ty = fmaf(h.f[3], x, fmaf(h.f[4], y, h.f[5]));
    float ox = -0.0f, oy = -0.0f;          // -0 + v = v for every v (IEEE): the first variation's add folds away (the CPU model of the tests starts at -0 as well)
    spec_variations<I, 0>(xf, h.w0, tx, ty, ox, oy, r);
    if constexpr (kSpecPost[I] != 0) {
        const float qx = fmaf(h.f[6], ox, fmaf(h.f[7], oy, h.f[8]));
        const float qy = fmaf(h.f[9], ox, fmaf(h.f[10], oy, h.f[11]));
        ox = qx; oy = qy;
    }
    const float csp = h.f[13];
    c = fmaf(c, 1.0f - csp, h.f[12] * csp);
    x = ox; y = oy;
}

// Kernels of few xforms keep EVERY xform's record in scalar registers for the whole launch (a slot's parameter block
// does not change during a launch): the round then neither computes the next record's address nor loads it (three
// scalar ALU instructions, up to five s_load and their share of the round's lgkmcnt waits — the scalar unit is on this
// kernel's critical path).  Nine registers per xform (pre affine, colour, speed, first weight; six more with a post affine).
#ifndef FL_RESIDENT_MAX_XF
#define FL_RESIDENT_MAX_XF 4
#endif
__host__ __device__ constexpr int spec_resident_sgprs()
{
    int n = 0;
    for (int i = 0; i < FL_SPEC_NXF; ++i) n += 9 + (kSpecPost[i] != 0 ? 6 : 0);
    return n;
}
constexpr bool kSpecResident = FL_SPEC_NXF <= FL_RESIDENT_MAX_XF && spec_resident_sgprs() <= 48;
// ... and what a round would otherwise rebuild from those scalars with VECTOR instructions, every round, lives in vector registers
// for the launch: a VALU instruction reads at most one scalar register, so `fma(a, x, fma(b, y, c))` with a, b, c in scalar
// registers costs a v_mov for c, and the colour blend `fma(col, 1 - speed, colour * speed)` computes its two wave-uniform
// operands with a v_sub, a v_mov and a v_mul (there is no scalar float ALU): 3 + 2 + 2 of a small flame's ~74 vector instructions
// per round.  Budget FL_HOIST_BUDGET registers (the 1536-slot geometry runs six waves per SIMD: 80 registers): the colour products
// first (3 instructions per register), the camera's offsets, then the affine offsets if they fit.  Same operations on the same
// values, done once: bit-identical.
#ifndef FL_HOIST_BUDGET
#define FL_HOIST_BUDGET 12
#endif
__host__ __device__ constexpr int spec_npost() { int n = 0; for (int i = 0; i < FL_SPEC_NXF; ++i) n += kSpecPost[i] != 0 ? 1 : 0; return n; }
constexpr bool kHoistCol = kSpecResident && FL_SPEC_NXF + 2 <= FL_HOIST_BUDGET;
constexpr bool kHoistAff = kHoistCol && 3 * FL_SPEC_NXF + 2 <= FL_HOIST_BUDGET;
constexpr bool kHoistPost = kHoistAff && 3 * FL_SPEC_NXF + 2 + 2 * spec_npost() <= FL_HOIST_BUDGET;
struct XfVec { float xo, yo, cprod, pxo, pyo; };
// spec_apply_xf for a resident record: h.f[13] holds 1 - colour speed, v the vector-register copies
// Kernels whose records are fetched per round (more than FL_RESIDENT_MAX_XF xforms) cannot keep every xform's operands in registers; they keep
// them in a 16-byte-per-xform LDS table {x offset, y offset, 1 - colour speed, colour * speed}, filled once per launch, and read the entry of the
// NEXT round's xform (chosen a round ahead) behind the swap: one address instruction and one ds_read_b128 per round instead of the two v_mov of
// the affine and the v_sub + v_mov + v_mul of the colour blend.
constexpr bool kTab = !kSpecResident && FL_HOIST_BUDGET >= 12 && FL_SPEC_NXF * 16 <= FL_XTAB_BYTES;
template <int I, class Extra>
__device__ __forceinline__ void spec_apply_xf_tab(const XfHead &h, const float4 &t, const float *__restrict__ xf,
                                                  float &x, float &y, float &c, mwc_t &r, const XfTail *tl, Extra &&extra)
{
    extra();          // (the previous round's plot, in the xform's own block: two independent chains for the scheduler; see iter_body)
    float tx = fmaf(h.f[0], x, fmaf(h.f[1], y, t.x));
    float ty = fmaf(h.f[3], x, fmaf(h.f[4], y, t.y));
    float ox = -0.0f, oy = -0.0f;
    spec_variations<I, 0>(xf, h.w0, tx, ty, ox, oy, r, tl);
    if constexpr (kSpecPost[I] != 0) {
        const float qx = fmaf(h.f[6], ox, fmaf(h.f[7], oy, h.f[8]));
        const float qy = fmaf(h.f[9], ox, fmaf(h.f[10], oy, h.f[11]));
        ox = qx; oy = qy;
    }
    c = fmaf(c, t.z, t.w);
    asm volatile("" : "+v"(c));
    x = ox; y = oy;
}
template <int LO, int HI, class Extra>
__device__ __forceinline__ void spec_dispatch_tab(int k, const XfHead &h, const float4 &t, const float *__restrict__ xf,
                                                  float &x, float &y, float &c, mwc_t &r, const XfTail *tl, Extra &&extra)
{
    if constexpr (HI - LO == 1) spec_apply_xf_tab<LO>(h, t, xf, x, y, c, r, tl, extra);
    else {
        constexpr int MID = (LO + HI) / 2;
        if (k < MID) spec_dispatch_tab<LO, MID>(k, h, t, xf, x, y, c, r, tl, extra);
        else spec_dispatch_tab<MID, HI>(k, h, t, xf, x, y, c, r, tl, extra);
    }
}
// The final xform's record is constant for the slot as well: its operands are held the same way (three registers, five with a post affine).
constexpr bool kHoistFinal = FL_SPEC_FINAL != 0 && FL_HOIST_BUDGET >= 7;
template <int I, bool COL = kHoistCol, bool AFF = kHoistAff, bool POST = kHoistPost>
__device__ __forceinline__ void spec_apply_xf_res(const XfHead &h, const XfVec &v, const float *__restrict__ xf,
                                                  float &x, float &y, float &c, mwc_t &r, const XfTail *tl = nullptr)
{
    float tx = fmaf(h.f[0], x, fmaf(h.f[1], y, AFF ? v.xo : h.f[2]));
    float ty = fmaf(h.f[3], x, fmaf(h.f[4], y, AFF ? v.yo : h.f[5]));
    float ox = -0.0f, oy = -0.0f;
    spec_variations<I, 0>(xf, h.w0, tx, ty, ox, oy, r, tl);
    if constexpr (kSpecPost[I] != 0) {
        const float qx = fmaf(h.f[6], ox, fmaf(h.f[7], oy, POST ? v.pxo : h.f[8]));
        const float qy = fmaf(h.f[9], ox, fmaf(h.f[10], oy, POST ? v.pyo : h.f[11]));
        ox = qx; oy = qy;
    }
    // (the empty asm keeps the blend in its xform's arm: merged into one v_fmac behind the arms it needs a v_mov of the product in each)
    if constexpr (COL) { c = fmaf(c, h.f[13], v.cprod); asm volatile("" : "+v"(c)); }
    else { const float csp = h.f[13]; c = fmaf(c, 1.0f - csp, h.f[12] * csp); }
    x = ox; y = oy;
}
template <int LO, int HI, class Extra>
__device__ __forceinline__ void spec_dispatch_res(int k, const XfHead (&heads)[FL_SPEC_NXF], const XfVec (&hv)[FL_SPEC_NXF], const float *__restrict__ xf0, int xf_stride,
                                                  float &x, float &y, float &c, mwc_t &r, const XfTail (&tails)[FL_SPEC_NXF], Extra &&extra)
{
    if constexpr (HI - LO == 1) { extra(); spec_apply_xf_res<LO>(heads[LO], hv[LO], xf0 + LO * xf_stride, x, y, c, r, FL_EARLY_TAIL ? &tails[LO] : nullptr); }
    else {
        constexpr int MID = (LO + HI) / 2;
        if (k < MID) spec_dispatch_res<LO, MID>(k, heads, hv, xf0, xf_stride, x, y, c, r, tails, extra);
        else spec_dispatch_res<MID, HI>(k, heads, hv, xf0, xf_stride, x, y, c, r, tails, extra);
    }
}

// wave-uniform dispatch over the selectable xforms [LO, HI): a binary tree of scalar compares
template <int LO, int HI, class Extra>
__device__ __forceinline__ void spec_dispatch(int k, const XfHead &h, const float *__restrict__ xf,
                                              float &x, float &y, float &c, mwc_t &r, Extra &&extra)
{
    if constexpr (HI - LO == 1) { extra(); spec_apply_xf<LO>(h, xf, x, y, c, r); }
    else {
        constexpr int MID = (LO + HI) / 2;
        if (k < MID) spec_dispatch<LO, MID>(k, h, xf, x, y, c, r, extra);
        else spec_dispatch<MID, HI>(k, h, xf, x, y, c, r, extra);
    }
}
#endif

// cuburn/code/iter.py:366-406: if the cell had reached 512 hits, swap it with zero and add its
// unpacked contents (weighted by the hot-pixel multiplier) to the float accumulator.
__device__ __forceinline__ void drain_if_full(bool ok, u64 old, uint32_t gi, float mult,
                                              u64 *__restrict__ atom, float *__restrict__ out4, uint32_t &n_spill)
{
    if (ok && (uint32_t)(old >> 32) >= (256u << 23)) {
        const u64 cur = __hip_atomic_exchange(atom + gi, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(cur >> 32) != 0u) {
            float yf, uf, vf, df;
            unpack_cell(cur, yf, uf, vf, df);
            const float m255 = mult * FL_INV255;
            float *o = out4 + 4 * (size_t)gi;
            __hip_atomic_fetch_add(o + 0, yf * m255, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(o + 1, uf * m255, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(o + 2, vf * m255, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(o + 3, df * mult, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++n_spill;
        }
    }
}

__device__ __forceinline__ void reseed(float &x, float &y, float &c, mwc_t &r) {
    x = mwc_next_11(r); y = mwc_next_11(r); c = mwc_next_01(r);
}

// Geometry of the binned accumulate (FL_ACCUM_BINNED): the image is cut into 128x64-pixel tiles
// ("bins"); a sample record is 21 bits {row in tile 6, column 7, palette column 8}; a staged
// record carries its bin number (11 bits) above that.
struct BinGeom {
    uint32_t tiles_x;        // tiles per row
    uint32_t nbins;          // number of tiles B (<= FL_MAX_BINS)
    uint32_t rounds;         // R: write-enabled rounds per batch
    uint32_t nbatch_total;   // batches of this launch = nslots * batches per slot
    uint32_t sub_log2;       // 0: a temporal sample per workgroup; 1 (8 waves) / 2 (16 waves): every four waves walk their own sample (2 or 4 per slot), see iter_body
};
// log: per batch a region of sorted records — R * NT 32-bit words (256x64 tiles), or fl_pack3_words(R * NT) 64-bit words of three
// records each (128x64 tiles, FL_LOG_PACK3); dir: [B][nbatch_total] (first record << 16) | count.
// They are separate __restrict__ kernel arguments: as members of the struct the compiler has to
// assume their stores may alias the parameter block, and every (wave-uniform) parameter load in
// the loop stops being an s_load.

// Inclusive add-scan over the 64 lanes with DPP (pure VALU: row_shr 1/2/4/8 inside each row of
// 16, then row_bcast:15 / row_bcast:31 carry the row totals) — ~10 instructions; the __shfl_up
// form is six ds_bpermute round trips through the LDS pipe.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// counters: [0] accepted, [1] out of frame, [2] dropped by hot-pixel roulette, [3] spills
// ACC: 0 = packed global atomics, 1 = binned (sample log, 128x64 tiles, tile number inside the staged
// record), 2 = none (measurement of the walk), 3 = binned for images with more than 2047 tiles
// (256x64 tiles, tile numbers staged in a separate 16-bit array)
template <int NW, bool COUNT, int ACC, bool SPEC>
__device__ __forceinline__ void
iter_body(unsigned char *smem, const int32_t *__restrict__ prog, const float *__restrict__ params,
          const u64 *__restrict__ palette, fl_mwc *__restrict__ rng, float4 *__restrict__ points,
          const uint32_t *__restrict__ hot, u64 *__restrict__ atom, float *__restrict__ out4,
          u64 *__restrict__ counters, uint32_t astride, uint32_t aheight,
          uint32_t round0, uint32_t nrounds, uint32_t fuse, BinGeom bg,
          uint32_t *__restrict__ bin_log, uint32_t *__restrict__ bin_dir)
{
    constexpr int NT = NW * 64;
    constexpr int SETS = NW == 4 ? FL_CNT_SETS : FL_CNT_SETS_BIG;
    constexpr bool BINNED = ACC == 1 || ACC == 3, WIDE = ACC == 3;
    constexpr bool PACK3 = FL_LOG_PACK3 != 0 && ACC == 1;              // 128x64 tiles: three 21-bit records per 64-bit log word
    constexpr uint32_t TWL = WIDE ? FL_TILE_W_WIDE_LOG2 : 7u;          // log2 of the tile width
    constexpr uint32_t PAY_BITS = TWL + FL_TILE_H_LOG2 + 8u;          // row | column | palette column
    // all LDS is carved from the dynamic region (16-byte aligned pieces)
    float (*swp)[3][NT] = reinterpret_cast<float (*)[3][NT]>(smem);                    // [2][3][NT]
    u64 *palrow = reinterpret_cast<u64 *>(smem + 2 * 3 * NT * 4);                      // [256]   (not binned)
    uint32_t *stage = reinterpret_cast<uint32_t *>(smem + 2 * 3 * NT * 4);             // [R*NT]  (binned)
    uint16_t *skey = reinterpret_cast<uint16_t *>(stage + bg.rounds * NT);             // [R*NT]  (wide only)
    uint32_t *cnt = stage + bg.rounds * NT + (WIDE ? bg.rounds * NT / 2 : 0);          // [B+1]
    // SETS sets of tile counters, lane l uses set l % SETS: the lanes of a wave hit few
    // distinct tiles, and LDS atomics on one address are applied one lane at a time — several sets
    // cut those chains (k_iter 0.80 -> 0.73 ms with two).  After the scan the scatter cursors of a
    // set take the place of its counters (zeroed again once the batch has been scattered).
    const uint32_t CNTW = (bg.nbins + 1 + 3) & ~3u;
    uint32_t *s_nvalid = cnt + SETS * CNTW;                                      // [4]
    uint32_t *tot = s_nvalid + 4;                                                      // [128] totals of 64-tile chunks
    // behind everything else (in FRONT it moved every other address off its immediate offsets: +6 registers): the per-xform operand
    // table of the per-genome kernels whose records are fetched per round, see kTab
    float4 *xtab = BINNED ? reinterpret_cast<float4 *>(tot + 128) : reinterpret_cast<float4 *>(palrow + FL_PAL_W);
    uint32_t *my_cnt = cnt + ((threadIdx.x & 63u) % SETS) * CNTW;

    const uint32_t tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    // Sub-blocks of four waves ("paired halves" of 8-wave workgroups, quarters of 16-wave ones; round 5): the reference walks 256
    // threads per temporal sample (cuburn/render.py:207, 1024 samples); an 8- or 16-wave workgroup per sample walks 512 or 1024,
    // i.e. two or four times the reference's un-plotted fuse rounds per frame.  With sub-blocks, waves 4q .. 4q + 3 walk sample
    // (NW / 4) * slot + q — walkers, RNG streams, point swap and parameter blocks exactly those of 1024 four-wave slots (same
    // results, bit for bit) — and the whole workgroup shares one sort batch, which is what the large workgroups are for.
    // (The wave number on the scalar side: the parameter block must stay a scalar base.)
#ifdef FL_RTC
    constexpr uint32_t kSubSpec = FL_SPEC_SUB_LOG2;
#else
    constexpr uint32_t kSubSpec = 0;
#endif
    const uint32_t sub_log2 = NW == 4 ? 0u : SPEC ? kSubSpec : bg.sub_log2;      // 0, or log2(NW / 4)
    const bool pair = sub_log2 != 0u;
    const uint32_t half = pair ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(w >> 2)) : 0u;      // which sub-block
    const uint32_t slot = blockIdx.x, ts = pair ? (slot << sub_log2) + half : slot, prow = slot * FL_PAL_H / gridDim.x;
#ifdef FL_RTC
    const int nxf = SPEC ? FL_SPEC_NXF : prog[1], has_final = SPEC ? FL_SPEC_FINAL : prog[2], pstride = SPEC ? FL_SPEC_PSTRIDE : prog[3];
    const int cdf_off = SPEC ? FL_SPEC_CDF_OFF : prog[4], xf_off = SPEC ? FL_SPEC_XF_OFF : prog[5];
    const int xf_stride = SPEC ? FL_SPEC_XF_STRIDE : prog[6], var_stride = SPEC ? FL_SPEC_VAR_STRIDE : prog[7];
#else
    const int nxf = prog[1], has_final = prog[2], pstride = prog[3], cdf_off = prog[4];
    const int xf_off = prog[5], xf_stride = prog[6], var_stride = prog[7];
#endif
    const float *__restrict__ P = params + (size_t)ts * pstride;

    if (!BINNED) for (int i = tid; i < FL_PAL_W; i += NT) palrow[i] = palette[prow * FL_PAL_W + i];
    if (BINNED) for (uint32_t i = tid; i < SETS * CNTW; i += NT) cnt[i] = 0;
    uint32_t batch_in_slot = 0;

    const size_t wi = (size_t)slot * NT + tid;
    mwc_t rctx = {rng[wi].mul, rng[wi].state, rng[wi].carry};
    float4 pt = points[wi];
    float x = pt.x, y = pt.y, color = pt.z;

    const float color_dither = 0.49f * mwc_next_11(rctx);                   // iter.py:185
    if (!isfinite(fabsf(x) + fabsf(y))) reseed(x, y, color, rctx);          // iter.py:209-216
    __syncthreads();
    // camera and final xform are constant for the slot
    const float cam0 = P[0], cam1 = P[1], cam3 = P[3], cam4 = P[4];
    float cam2 = P[2], cam5 = P[5];
#ifdef FL_RTC
    if constexpr (SPEC && FL_HOIST_BUDGET >= 2) asm volatile("" : "+v"(cam2), "+v"(cam5));      // the camera's offsets in vector registers (see kHoistCol)
#endif
    const float *__restrict__ xf_final = P + xf_off + nxf * xf_stride;

    // Cumulative xform densities of this slot's temporal sample (constant for the launch), one
    // per lane: the wave-uniform choice is then ONE vector compare + find-first-set instead
    // of a scalar compare chain.
    // Lane l holds the selector threshold of xform l: the xform is chosen when
    //     (float)sel * 2^-32 <= cdf[l]                                   (iter.py:260-272)
    // The left side is a non-decreasing function of the 32-bit selector, so the condition is
    // sel <= T_l with T_l the largest selector that satisfies it — found once per launch by
    // bisection on exactly that float expression.  The per-round choice is then ONE integer
    // compare against the (scalar) selector; lanes past the last density never match.
    const float cdf_lane = ((int)l < nxf - 1 && l < 63u) ? P[cdf_off + (int)l] : -1.0f;
    auto sel_ok = [&](uint32_t s) -> bool { return (float)s * (1.0f / 4294967296.0f) <= cdf_lane; };
    const bool thr_valid = sel_ok(0u);
    uint32_t thr = 0xffffffffu;
    if (!sel_ok(thr)) {
        uint32_t lo = 0u, hi = 0xffffffffu;             // sel_ok(lo) holds (where thr_valid), sel_ok(hi) does not
        for (int it = 0; it < 32; ++it) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (sel_ok(mid)) lo = mid; else hi = mid;
        }
        thr = lo;
    }
    // swap destinations of this and the next two rounds (phase = round % 3), rotated every round
    auto swap_dest = [&](const uint32_t phase) -> uint32_t {
        return pair ? (w & ~3u) * 64u + shuffle_dest<4>(w & 3u, l, phase) : shuffle_dest<NW>(w, l, phase);      // sub-blocks: within the four waves
    };
    uint32_t rot0 = swap_dest(round0 % 3u), rot1 = swap_dest((round0 + 1u) % 3u), rot2 = swap_dest((round0 + 2u) % 3u);
    uint32_t n_acc = 0, n_oob = 0, n_drop = 0, n_spill = 0;
    bool pend_ok = false; uint32_t pend_gi = 0; float pend_mult = 1.0f; u64 pend_old = 0;

    // Wave-coherent xform choice: lane 0's draw (iter.py:260-272 uses a shared cosel[]).  The
    // selector of round r+1 is drawn at the top of round r, so the record it picks can be
    // fetched a whole round before it is needed.
    const unsigned long long valid_lanes = __ballot(thr_valid);            // scalar, once per launch
    auto choose = [&](uint32_t sel) -> int {
        // smallest i with sel <= T_i; nxf - 1 if there is none.  (The valid lanes are ANDed in on the scalar side:
        // `__ballot(thr_valid && ...)` made the compiler rebuild the predicate in a VGPR: 3 vector instructions, now 1.)
        const unsigned long long le = (__ballot(sel <= thr) & valid_lanes) | (1ull << 63);
        return min((int)__builtin_ctzll(le), nxf - 1);
    };
    uint32_t sel_next = __builtin_amdgcn_readfirstlane(mwc_next(rctx));
    int k_next = choose(sel_next);
    const float *__restrict__ xf_next = P + xf_off + k_next * xf_stride;
#ifdef FL_RTC
    constexpr bool RESIDENT = SPEC && kSpecResident;
    XfHead heads[FL_SPEC_NXF];
    XfVec hv[FL_SPEC_NXF] = {};
    XfTail tails[FL_SPEC_NXF] = {};        // resident kernels: the words behind each head that the record's variations read stay in scalar registers too (the others are dead code)
    if constexpr (RESIDENT) {
#pragma unroll
        for (int i = 0; i < FL_SPEC_NXF; ++i) {
            heads[i] = load_head(P + xf_off + i * xf_stride);
            if constexpr (FL_EARLY_TAIL) {
#pragma unroll
                for (int k = 0; k < kTailWords; ++k) tails[i].w[k] = (P + xf_off + i * xf_stride)[kTailFirst + k];
            }
            if constexpr (kHoistCol) {
                const float csp = heads[i].f[13];
                hv[i].cprod = heads[i].f[12] * csp;
                asm volatile("" : "+v"(hv[i].cprod));
                heads[i].f[13] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f - csp)));
            }
            if constexpr (kHoistAff) { hv[i].xo = heads[i].f[2]; hv[i].yo = heads[i].f[5]; asm volatile("" : "+v"(hv[i].xo), "+v"(hv[i].yo)); }
            if constexpr (kHoistPost) { hv[i].pxo = heads[i].f[8]; hv[i].pyo = heads[i].f[11]; asm volatile("" : "+v"(hv[i].pxo), "+v"(hv[i].pyo)); }
        }
    }
    XfHead hfin_res = {};
    XfTail tfin = {};                      // (the final xform's record is constant for the slot: its tail words in use stay in registers as well)
    XfVec vfin = {};
    if constexpr (SPEC && kHoistFinal) {
        hfin_res = load_head(xf_final);
        if constexpr (FL_EARLY_TAIL) {
#pragma unroll
            for (int k = 0; k < kTailWords; ++k) tfin.w[k] = xf_final[kTailFirst + k];
        }
        const float csp = hfin_res.f[13];
        vfin.cprod = hfin_res.f[12] * csp;
        vfin.xo = hfin_res.f[2]; vfin.yo = hfin_res.f[5];
        asm volatile("" : "+v"(vfin.cprod), "+v"(vfin.xo), "+v"(vfin.yo));
        hfin_res.f[13] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f - csp)));
        if constexpr (kSpecPost[FL_SPEC_NXF] != 0) { vfin.pxo = hfin_res.f[8]; vfin.pyo = hfin_res.f[11]; asm volatile("" : "+v"(vfin.pxo), "+v"(vfin.pyo)); }
    }
#else
    constexpr bool RESIDENT = false;
#endif
    XfHead hnext = RESIDENT ? XfHead{} : load_head(xf_next);
#ifdef FL_RTC
    XfTail tail_next = {};
    if constexpr (SPEC && kTab && FL_EARLY_TAIL) {
#pragma unroll
        for (int i = 0; i < kTailWords; ++i) tail_next.w[i] = xf_next[kTailFirst + i];
    }
#endif
    float4 tcur = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#ifdef FL_RTC
    if constexpr (SPEC && kTab) {
        const uint32_t th = pair ? tid & 255u : tid;           // (paired: a table per half, filled from the half's own block)
        if ((int)th < FL_SPEC_NXF) {
            const float *__restrict__ rr = P + xf_off + (int)th * xf_stride;
            const float csp = rr[13];
            xtab[half * 16u + th] = make_float4(rr[2], rr[5], 1.0f - csp, rr[12] * csp);
        }
        __syncthreads();
        tcur = xtab[half * 16u + k_next];
    }
#endif

    // One round of the walk: reseed bad points, apply the chosen xform, swap walkers between waves.
    uint32_t par = 0;                                   // parity of the round: which of the two swap buffers
    auto advance = [&](const uint32_t dst, auto &&extra) __attribute__((always_inline)) {
        // Wave priority: the walk — everything up to the swap's barrier and the reads behind it — runs at a raised
        // priority, the plotting that follows at the normal one.  A SIMD holds one wave of each of six workgroups; the
        // wave whose three siblings (on the other SIMDs) already wait at the round's barrier should not queue behind
        // waves that are packing records: k_iter_spec 672 -> 632 us ALONE (level 1, 2 or 3: the same; lowering it later, or
        // raising it only for the swap, gains less: profiles/r03_wave_priority.txt).  In the two-lane frame loop the other
        // lane's kernels already fill those issue slots: the frame time does not move; a single frame gains the 6 %.
#if FL_ITER_PRIO
        __builtin_amdgcn_s_setprio(FL_ITER_PRIO);
#endif
        if (!isfinite(fabsf(x) + fabsf(y))) reseed(x, y, color, rctx);      // iter.py:225-229

        const int k_cur = k_next;
        const float *__restrict__ xf_cur = xf_next;
#ifdef FL_RTC
        XfTail tail = tail_next;          // (FL_EARLY_TAIL: requested a round ahead, with the head)
#endif
        sel_next = __builtin_amdgcn_readfirstlane(mwc_next(rctx));
        k_next = choose(sel_next);
        xf_next = P + xf_off + k_next * xf_stride;
        // The next record is requested AFTER this round's xform has used the current one: it lands in the
        // same scalar registers (no second set and no nine register copies every round — the scalar unit is
        // on this kernel's critical path) and still has the swap, the barrier and the rest of the round to
        // arrive: k_iter 0.717 -> 0.70 ms, the interpreter kernel -15 %.
#ifdef FL_RTC
        if constexpr (RESIDENT) spec_dispatch_res<0, FL_SPEC_NXF>(k_cur, heads, hv, P + xf_off, xf_stride, x, y, color, rctx, tails, extra);
        else if constexpr (SPEC && kTab) spec_dispatch_tab<0, FL_SPEC_NXF>(k_cur, hnext, tcur, xf_cur, x, y, color, rctx, FL_EARLY_TAIL ? &tail : nullptr, extra);
        else if constexpr (SPEC) spec_dispatch<0, FL_SPEC_NXF>(k_cur, hnext, xf_cur, x, y, color, rctx, extra);
        else
#endif
        { extra(); apply_xf(hnext, xf_cur, var_stride, x, y, color, rctx); }
        if constexpr (!RESIDENT) hnext = load_head(xf_next);
#ifdef FL_RTC
        if constexpr (SPEC && kTab && FL_EARLY_TAIL) {
#pragma unroll
            for (int i = 0; i < kTailWords; ++i) tail_next.w[i] = xf_next[kTailFirst + i];
        }
#endif
        (void)k_cur; (void)xf_cur;

        // rotate walkers between waves (iter.py:274-294), double-buffered by round parity
        {
#if defined(FL_X_NOSWAP)            /* timing experiments only (wrong results): no point swap at all ... */
            (void)dst;
#elif defined(FL_X_HALF_BARRIERS)   /* ... or a barrier every other round */
            swp[par][0][dst] = x; swp[par][1][dst] = y; swp[par][2][dst] = color;
            if (par) __syncthreads();
            x = swp[par][0][tid]; y = swp[par][1][tid]; color = swp[par][2][tid];
#else
            swp[par][0][dst] = x; swp[par][1][dst] = y; swp[par][2][dst] = color;
            __syncthreads();
            x = swp[par][0][tid]; y = swp[par][1][tid]; color = swp[par][2][tid];
#endif
#ifdef FL_RTC
            if constexpr (SPEC && kTab) tcur = xtab[half * 16u + k_next];          // the next round's operands: on their way while this round plots
#endif
#if FL_ITER_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            par ^= 1u;
        }
    };
    // The loop is split by what a round does besides walking — nothing (the fuse rounds, iter.py:298-300),
    // or plotting, in batches of bg.rounds rounds that end with the tile sort — so that a round's loop
    // bookkeeping is ONE counter: the scalar unit is on this kernel's critical path.
    const uint32_t nfuse = min(fuse, nrounds);
#ifdef FL_RTC
    constexpr bool SPLIT_FUSE = !SPEC || FL_SPEC_NXF <= FL_SPLIT_MAX_XF;
#else
    constexpr bool SPLIT_FUSE = true;
#endif
    // (a kernel with many heavy xforms keeps ONE copy of the walk: there the fuse rounds run through the
    // plotting loop with the plot skipped — a second copy of twelve inlined xforms cost cfg5 7 %)
    // The swap destination cycles through three values (phase = round % 3).  Rotating three registers costs three
    // v_mov per round; small kernels instead carry THREE copies of the round, one per destination, entered at the
    // current phase (ROT3; kernels of many heavy xforms keep one copy and rotate).
    constexpr bool ROT3 = SPLIT_FUSE && FL_ITER_ROT3;
    uint32_t ph = 0;                                    // which of rot0 / rot1 / rot2 the next round uses (ROT3)
    auto next_dst = [&]() __attribute__((always_inline)) -> uint32_t {      // one copy: rotate
        const uint32_t dst = rot0;
        rot0 = rot1; rot1 = rot2; rot2 = dst;
        return dst;
    };
    // run `body(dst, i)` for i = from .. to - 1 with the destinations in cycle
    auto rounds = [&](uint32_t from, uint32_t to, auto body) __attribute__((always_inline)) {
        if constexpr (ROT3) {
            uint32_t i = from;
            while (i < to) {
                if (ph == 0u) { body(rot0, i); ph = 1u; if (++i == to) break; }
                if (ph == 1u) { body(rot1, i); ph = 2u; if (++i == to) break; }
                body(rot2, i); ph = 0u; ++i;
            }
        } else {
            for (uint32_t i = from; i < to; ++i) body(next_dst(), i);
        }
    };
#ifdef FL_RTC
    constexpr bool MERGE = BINNED && SPEC && SPLIT_FUSE && FL_SPEC_NXF <= FL_ITER_MERGE_MAX_XF;
#else
    constexpr bool MERGE = false;
#endif
    // The plot of a round, in two parts: the final xform (which may draw random numbers) ...
    auto plot_head = [&](float &cx, float &cy, float &cf) __attribute__((always_inline)) {
        float fx = x, fy = y, fc = color;
#ifdef FL_RTC
        if constexpr (SPEC) {
            if constexpr (kHoistFinal) spec_apply_xf_res<FL_SPEC_NXF, true, true, true>(hfin_res, vfin, xf_final, fx, fy, fc, rctx, FL_EARLY_TAIL ? &tfin : nullptr);
            else if constexpr (FL_SPEC_FINAL != 0) { const XfHead hfin = load_head(xf_final); spec_apply_xf<FL_SPEC_NXF>(hfin, xf_final, fx, fy, fc, rctx); }
        } else
#endif
        if (has_final) { const XfHead hfin = load_head(xf_final); apply_xf(hfin, xf_final, var_stride, fx, fy, fc, rctx); }   // iter.py:302-307
        // the camera and the colour's scale here too: the point's registers are free before the next round's reseed can
        // write them (kept for the second part, the MERGE loop below paid three register copies per round)
        cx = fmaf(cam0, fx, fmaf(cam1, fy, cam2));                          // iter.py:306-309
        cy = fmaf(cam3, fx, fmaf(cam4, fy, cam5));
        // iter.py:346-348; rint of a value in [0, 255] = the low mantissa bits of (value + 2^23): the float adder rounds to nearest even
        cf = fminf(fmaxf(fmaf(fc, 255.0f, color_dither), 0.0f), 255.0f) + 8388608.0f;
        if constexpr (MERGE) asm volatile("" : "+v"(cx), "+v"(cy), "+v"(cf));      // (here, not sunk behind the reseed)
    };
    // ... and the cell and the record or the add (no random numbers in the binned back-end)
    auto plot_rest = [&](const float cx, const float cy, const float cf, const uint32_t staged) __attribute__((always_inline)) {
        // iter.py:313-317: round to nearest even, reject outside [0, astride) x [0, aheight).
        // Adding 1.5 * 2^23 rounds to the nearest integer (ties to even) in the float adder and leaves
        // that integer in the low mantissa bits: for every finite or non-finite cx the difference
        // to the bit pattern of 1.5 * 2^23 is rint(cx) if that lies in [0, 2^22), and a value >= 2^22
        // (as unsigned) otherwise — negative, huge, infinite and NaN inputs all fail the range test.
        const uint32_t ix = __float_as_uint(cx + 12582912.0f) - 0x4b400000u;
        const uint32_t iy = __float_as_uint(cy + 12582912.0f) - 0x4b400000u;
        bool ok = (ix < astride) & (iy < aheight);
        // (binned mode: a rejected sample only needs its tile number forced to "none" below, its
        // coordinate bits are never looked at)
        if (COUNT) n_oob += !ok;
        const uint32_t gi = ok ? iy * astride + ix : 0u;

        float mult = 1.0f;
        if (ok && !BINNED) {                                                // iter.py:319-329
            const uint32_t flag = (hot[gi >> 4] >> ((gi & 15u) << 1)) & 3u;
            if (flag) {
                mult = hot_mult(flag);
                if (mwc_next_01(rctx) > frcp(mult)) { ok = false; if (COUNT) ++n_drop; }
            }
        }
        const uint32_t cbits = __float_as_uint(cf);
        const int ci = (int)((cbits - 0x4b000000u) & 0xffu);
        const u64 val = BINNED ? 0ull : palrow[ci];                         // iter.py:351

        // Every add returns the previous cell value, but the value is only looked at one
        // round later (pend_*), so its latency hides under the next round's work.  A cell seen
        // at >= 512 hits is drained into the float accumulator before its 10-bit count can
        // wrap.  (The reference checks 3 % of warp-rounds synchronously, iter.py:361-406; with
        // wave-coherent hits that leaves a real chance of wrapping on concentrated flames.)
        if (ACC == 0) {
            drain_if_full(pend_ok, pend_old, pend_gi, pend_mult, atom, out4, n_spill);
            pend_ok = ok; pend_gi = gi; pend_mult = mult;
            if (ok) pend_old = __hip_atomic_fetch_add(atom + gi, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (BINNED) {
            // stage the record, count its bin; every R rounds the batch is sorted by bin in LDS
            // and written to this slot's private region of the sample log (no global atomics)
            const uint32_t bin = ok ? __umul24(iy >> FL_TILE_H_LOG2, bg.tiles_x) + (ix >> TWL) : bg.nbins;
            // {bin | row | column} in three bytes, then ONE v_perm_b32 shifts them up a byte and drops the palette column — byte 0 of
            // the rounded colour's bit pattern — underneath: two ands, two shift-ors and the permute, where masking and shifting every
            // field into place took seven instructions + the colour's own mask
            const uint32_t t1 = ((iy & (FL_TILE_H - 1u)) << TWL) | (ix & ((1u << TWL) - 1u));
            const uint32_t t2 = WIDE ? t1 : (bin << (TWL + FL_TILE_H_LOG2)) | t1;
            const uint32_t rec = __builtin_amdgcn_perm(t2, cbits, 0x06050400u);
            if (WIDE) skey[staged * NT + tid] = (uint16_t)bin;
            stage[staged * NT + tid] = rec;
#ifndef FL_X_NO_CNT         /* timing experiment (wrong results): no count per record */
            __hip_atomic_fetch_add(my_cnt + bin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        } else {
            // measurement mode: everything but the accumulate (ceiling of the walk itself)
            pend_old += ok ? val + gi : 0ull;
        }
        if (COUNT) n_acc += ok;
    };
    auto no_extra = []() __attribute__((always_inline)) {};
    // MERGE: the walk of round k + 1 does not depend on the plot of round k past its final xform — the camera, the
    // cell, the record, the count.  Placed in the chosen xform's own block that work gives the scheduler a second,
    // independent chain to fill the first one's latencies with (the kernel waits on dependent issue at four waves per
    // SIMD, not on throughput).  Same order of everything a lane draws and writes: A1 P1 A2 P2 ... An Pn.
    if (SPLIT_FUSE) rounds(0u, MERGE && nfuse < nrounds ? nfuse + 1u : nfuse, [&](uint32_t dst, uint32_t) __attribute__((always_inline)) { advance(dst, no_extra); });
    uint32_t fuse_left = SPLIT_FUSE ? 0u : nfuse;
    for (uint32_t rd = SPLIT_FUSE ? nfuse : 0u; rd < nrounds;) {
    const uint32_t blen = fuse_left ? fuse_left : BINNED ? min(bg.rounds, nrounds - rd) : nrounds - rd;
    const bool plotting = fuse_left == 0u;
    fuse_left = 0u;
    if constexpr (MERGE) {
        // the walk of this batch's first round has run; its last round's walk belongs to the next batch
        const uint32_t last = rd + blen >= nrounds ? 1u : 0u;
        rounds(0u, blen - last, [&](const uint32_t dst, const uint32_t staged) __attribute__((always_inline)) {
            float fx, fy, fc;
            plot_head(fx, fy, fc);          // (camera-space x, y and the scaled colour)
            advance(dst, [&]() __attribute__((always_inline)) { plot_rest(fx, fy, fc, staged); });
        });
        if (last) { float fx, fy, fc; plot_head(fx, fy, fc); plot_rest(fx, fy, fc, blen - 1u); }
    } else
    rounds(0u, blen, [&](const uint32_t dst, const uint32_t staged) __attribute__((always_inline)) {
        advance(dst, no_extra);
        if (!SPLIT_FUSE && !plotting) return;
        float fx, fy, fc;
        plot_head(fx, fy, fc);
        plot_rest(fx, fy, fc, staged);
    });
    rd += blen;
#ifdef FL_X_SKIP_SORT      /* timing experiment (wrong results): no batch epilogue at all */
    if (false) {
#else
    if (BINNED && plotting) {
#endif
            // The batch epilogue, compiled twice: for the full batch — every "is there a record" test folds away and the scatter
            // is straight-line code — and for a launch's short last batch.
            auto sort_batch = [&](const uint32_t staged) __attribute__((always_inline)) {
#if FL_SORT_LOCAL_TID
                // The thread number behind an empty asm: the addresses the epilogue derives from it are then computed here, once
                // per batch, instead of being held in ~15 registers across the whole kernel (the compiler hoists them as
                // loop invariants; the epilogue is where the register count peaks).
                uint32_t tid_local = threadIdx.x;
                asm volatile("" : "+v"(tid_local));
                const uint32_t tid = tid_local, l = tid & 63u;
                const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
#endif
                // slot-major: the batches of a slot are neighbours in the log and the directory, so a range
                // of batch ids covers few slots, i.e. few palette rows (binned.hip stages them in LDS)
                const uint32_t batch_id = slot * (bg.nbatch_total / gridDim.x) + batch_in_slot;
                __syncthreads();
                // Each thread takes its own staged records into registers (while wave 0 scans the
                // tile counts), so that the scatter below can sort the batch IN PLACE: no second
                // R*NT buffer, which is what limits the workgroups per CU.
                uint32_t r2[FL_BIN_R_MAX], k2[FL_BIN_R_MAX];       // payload, tile (0xffffffff: no record)
#pragma unroll
                for (int q = 0; q < FL_BIN_R_MAX; ++q) {
                    const bool have = (uint32_t)q < staged;
                    r2[q] = have ? stage[q * NT + tid] : 0u;
                    k2[q] = !have ? 0xffffffffu : WIDE ? (uint32_t)skey[q * NT + tid] : r2[q] >> PAY_BITS;
                }
                // Exclusive scan of the tile counts -> scatter cursors and directory entries, by all
                // waves: chunks of 64 tiles are scanned where they lie (wave w takes chunks w, w + NW,
                // ...), then every wave adds the prefix of the chunk totals to its own chunks.  (One
                // wave walking all 2109 tiles of an 8K image kept the other fifteen waiting for 15 %
                // of the kernel.)
                const uint32_t nchunk = (bg.nbins + 64u) >> 6;          // tiles 0..nbins, the last one = "no record"
#ifdef FL_X_NO_SCAN         /* timing experiment (wrong results) */
                if (true) { if (tid == 0) *s_nvalid = staged * NT; } else
#endif
                if (nchunk <= FL_SCAN_SERIAL_MAX) { // few tiles (1080p: 4 chunks): one wave is quicker than a second barrier
                    if (w == 0) {
                        uint32_t running = 0;
                        for (uint32_t c0 = 0; c0 <= bg.nbins; c0 += 64) {
                            const uint32_t b = c0 + l;
                            uint32_t vs[SETS], v = 0;
#pragma unroll
                            for (int t = 0; t < SETS; ++t) { vs[t] = b <= bg.nbins ? cnt[t * CNTW + b] : 0u; v += vs[t]; }
                            const uint32_t incl = wave_incl_scan(v, l);
                            const uint32_t excl = incl - v + running;
                            if (b <= bg.nbins) {
                                uint32_t at = excl;
#pragma unroll
                                for (int t = 0; t < SETS; ++t) { cnt[t * CNTW + b] = at; at += vs[t]; }
                            }
                            if (b < bg.nbins) bin_dir[(size_t)b * bg.nbatch_total + batch_id] = (excl << 16) | v;      // (plain stores: non-temporal ones here cost k_iter +60 %)
                            if (b == bg.nbins) *s_nvalid = excl;
                            running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        }
                    }
                } else {
                for (uint32_t c = w; c < nchunk; c += NW) {
                    const uint32_t b = c * 64u + l;
                    uint32_t v = 0;
#pragma unroll
                    for (int t = 0; t < SETS; ++t) v += b <= bg.nbins ? cnt[t * CNTW + b] : 0u;
                    const uint32_t incl = wave_incl_scan(v, l);
                    if (l == 63u) tot[c] = incl;
                }
                __syncthreads();
                {
                    const uint32_t t0 = l < nchunk ? tot[l] : 0u, t1 = 64u + l < nchunk ? tot[64u + l] : 0u;   // nchunk <= 128
                    const uint32_t i0 = wave_incl_scan(t0, l);
                    const uint32_t i1 = wave_incl_scan(t1, l) + (uint32_t)__builtin_amdgcn_readlane((int)i0, 63);
                    for (uint32_t cv = w; cv < nchunk; cv += NW) {
                        const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cv);
                        const uint32_t base = c < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)(i0 - t0), c)
                                                      : (uint32_t)__builtin_amdgcn_readlane((int)(i1 - t1), c - 64u);
                        const uint32_t b = c * 64u + l;
                        uint32_t vs[SETS], v = 0;                  // (the chunk is scanned again rather than stored in between)
#pragma unroll
                        for (int t = 0; t < SETS; ++t) { vs[t] = b <= bg.nbins ? cnt[t * CNTW + b] : 0u; v += vs[t]; }
                        const uint32_t excl = wave_incl_scan(v, l) - v + base;
                        if (b <= bg.nbins) {
                            uint32_t at = excl;
#pragma unroll
                            for (int t = 0; t < SETS; ++t) { cnt[t * CNTW + b] = at; at += vs[t]; }
                            if (b < bg.nbins) bin_dir[(size_t)b * bg.nbatch_total + batch_id] = (excl << 16) | v;      // (plain stores: non-temporal ones here cost k_iter +60 %)
                            else *s_nvalid = excl;
                        }
                    }
                }
                }
                __syncthreads();
                // scatter, four records per thread in flight (the returning LDS atomic is a
                // ~100-cycle round trip; one at a time this loop was a quarter of the kernel)
#ifndef FL_X_NO_SCATTER     /* timing experiment (wrong results) */
#pragma unroll
                for (int q0 = 0; q0 < FL_BIN_R_MAX; q0 += FL_SCATTER_DEPTH) {
                    if ((uint32_t)q0 >= staged) break;
                    uint32_t pos[FL_SCATTER_DEPTH];
#pragma unroll
                    for (int q = 0; q < FL_SCATTER_DEPTH; ++q)
                        if (k2[q0 + q] != 0xffffffffu)
#ifdef FL_X_SCATTER_READS   /* timing experiment (wrong results): a plain read of the cursor instead of the returning add */
                            pos[q] = (my_cnt[k2[q0 + q]] + (uint32_t)(q0 + q) * NT + tid) & (uint32_t)(FL_BIN_R_MAX * NT - 1);
#else
                            pos[q] = __hip_atomic_fetch_add(my_cnt + k2[q0 + q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
#pragma unroll
                    for (int q = 0; q < FL_SCATTER_DEPTH; ++q)
#if defined(FL_X_NO_SCAN)
                        if (k2[q0 + q] != 0xffffffffu) stage[pos[q] & (uint32_t)(FL_BIN_R_MAX * NT - 1)] = r2[q0 + q] & ((1u << PAY_BITS) - 1u);
#else
                        if (k2[q0 + q] != 0xffffffffu) stage[pos[q]] = r2[q0 + q] & ((1u << PAY_BITS) - 1u);
#endif
                }
#endif
                __syncthreads();
#ifdef FL_X_NO_PACK         /* timing experiment (wrong results) */
                const uint32_t nvalid = 0;
#else
                const uint32_t nvalid = *s_nvalid;
#endif
                if constexpr (PACK3) {
                    // three 21-bit records to an aligned 64-bit word (flame_device.h): the log is what the accumulate streams, and at
                    // 1080p it is bound by those bytes.  The slots behind the batch's last record hold whatever the LDS held (the
                    // accumulate takes only the slots its directory entries name; a stale word's high bits land in slots that are
                    // stale themselves).  12-byte stride per lane: conflict-free.
                    typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
                    u32x2_ *d2 = reinterpret_cast<u32x2_ *>(bin_log) + (size_t)batch_id * fl_pack3_words(bg.rounds * NT);
                    const uint32_t nwords = (nvalid + 2u) / 3u;
                    for (uint32_t i = tid; i < nwords; i += NT) {
                        const uint32_t a = stage[3u * i], b = stage[3u * i + 1u], c3 = stage[3u * i + 2u];
                        u32x2_ o;
                        o.x = a | (b << 21); o.y = (b >> 11) | (c3 << 10);
#if FL_LOG_NT      /* the log is written once and read by another kernel much later */
                        __builtin_nontemporal_store(o, d2 + i);
#else
                        d2[i] = o;
#endif
                    }
                } else {
                uint4 *dst = reinterpret_cast<uint4 *>(bin_log + (size_t)batch_id * bg.rounds * NT);
#if FL_LOG_NT
                {
                    typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
                    const u32x4_ *s4 = reinterpret_cast<const u32x4_ *>(stage);
                    u32x4_ *d4 = reinterpret_cast<u32x4_ *>(dst);
                    for (uint32_t i = tid; i * 4 < nvalid; i += NT) __builtin_nontemporal_store(s4[i], d4 + i);
                }
#else
                for (uint32_t i = tid; i * 4 < nvalid; i += NT) dst[i] = reinterpret_cast<const uint4 *>(stage)[i];
#endif
                }
                for (uint32_t i = tid; i < SETS * CNTW; i += NT) cnt[i] = 0;               // the cursors become counters again
                ++batch_in_slot;
                __syncthreads();
            };
#if FL_SORT_FULL_COPY
            if (blen == (uint32_t)FL_BIN_R_MAX) sort_batch((uint32_t)FL_BIN_R_MAX); else
#endif
            sort_batch(blen);
    }
    }

    if (ACC == 0) drain_if_full(pend_ok, pend_old, pend_gi, pend_mult, atom, out4, n_spill);
    else if (pend_old == 0x123456789abcdefull) atom[0] = pend_old;
    {   // (the walker's index again, behind an empty asm: two 64-bit addresses are not held across the kernel)
        uint32_t tid_end = threadIdx.x;
        asm volatile("" : "+v"(tid_end));
        const size_t wj = (size_t)slot * NT + tid_end;
        points[wj] = make_float4(x, y, color, 0.0f);                        // iter.py:414-416
        rng[wj].mul = rctx.mul; rng[wj].state = rctx.state; rng[wj].carry = rctx.carry;
    }

    if (COUNT) {
        atomicAdd(counters + 0, (u64)n_acc);
        atomicAdd(counters + 1, (u64)n_oob);
        atomicAdd(counters + 2, (u64)n_drop);
        atomicAdd(counters + 3, (u64)n_spill);
    }
}

#define FL_ITER_ARGS const int32_t *__restrict__ prog, const float *__restrict__ params, \
       const u64 *__restrict__ palette, fl_mwc *__restrict__ rng, float4 *__restrict__ points, \
       const uint32_t *__restrict__ hot, u64 *__restrict__ atom, float *__restrict__ out4, \
       u64 *__restrict__ counters, uint32_t astride, uint32_t aheight, \
       uint32_t round0, uint32_t nrounds, uint32_t fuse, BinGeom bg, \
       uint32_t *__restrict__ bin_log, uint32_t *__restrict__ bin_dir
#define FL_ITER_PASS prog, params, palette, rng, points, hot, atom, out4, counters, astride, aheight, round0, nrounds, fuse, bg, bin_log, bin_dir

#ifdef FL_RTC
// the kernel of ONE genome structure, walker geometry and accumulate mode (rtc.hip compiles it on
// first use and caches the code object)
#ifdef FL_X_WAVES_PER_EU    /* experiment: cap the vector registers (8: 64, 7: 72, 6: 80 ...) */
__attribute__((amdgpu_waves_per_eu(FL_X_WAVES_PER_EU, FL_X_WAVES_PER_EU)))
#endif
extern "C" __global__ void __launch_bounds__(FL_SPEC_NW * 64) k_iter_spec(FL_ITER_ARGS)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    iter_body<FL_SPEC_NW, FL_SPEC_COUNT != 0, FL_SPEC_ACC, true>(smem, FL_ITER_PASS);
}
#else
template <int NW, bool COUNT, int ACC>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 6 : NW == 8 ? 3 : 1) k_iter(FL_ITER_ARGS)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    iter_body<NW, COUNT, ACC, false>(smem, FL_ITER_PASS);
}

// Xform tap: apply xform `xfi` of temporal sample `ts` once to n independent points (one per
// thread, own RNG state).  For the per-variation parity tests.
__global__ void __launch_bounds__(256)
k_apply_xf_tap(const int32_t *__restrict__ prog, const float *__restrict__ params, uint32_t ts, int xfi,
               uint32_t n, float4 *__restrict__ pts, fl_mwc *__restrict__ rng)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float *__restrict__ P = params + (size_t)ts * prog[3];
    mwc_t r = {rng[i].mul, rng[i].state, rng[i].carry};
    float4 p = pts[i];
    const float *__restrict__ xf = P + prog[5] + xfi * prog[6];
    const XfHead h = load_head(xf);
    apply_xf(h, xf, prog[7], p.x, p.y, p.z, r);
    pts[i] = p;
    rng[i].mul = r.mul; rng[i].state = r.state; rng[i].carry = r.carry;
}

void launch_apply_xf_tap(hipStream_t st, const int32_t *prog, const float *params, uint32_t ts, int xfi,
                         uint32_t n, float4 *pts, fl_mwc *rng)
{
    hipLaunchKernelGGL(k_apply_xf_tap, dim3((n + 255) / 256), dim3(256), 0, st, prog, params, ts, xfi, n, pts, rng);
}

// Point-shuffle tap: one swap of the identity payload, for the bit-exact permutation test.
template <int NW>
__global__ void __launch_bounds__(NW * 64) k_shuffle_tap(uint32_t *out, uint32_t round)
{
    __shared__ uint32_t s[NW * 64];
    const uint32_t tid = threadIdx.x;
    s[shuffle_dest<NW>(tid >> 6, tid & 63, round % 3u)] = tid;
    __syncthreads();
    out[tid] = s[tid];
}

// cuburn/code/iter.py:420-544 flush_atom: drain packed cells into the float accumulator with
// the hot-flag weight that was in force while they filled, zero them, and recompute the 2-bit
// flags (16 pixels per u32 word at gi >> 4) from the accumulated density.
__global__ void __launch_bounds__(256)
k_flush(u64 *__restrict__ atom, float4 *__restrict__ out, uint32_t *__restrict__ hot, uint32_t nbins, int use_hot)
{
    const uint32_t gi = blockIdx.x * 256u + threadIdx.x;      // nbins is a multiple of 512
    if (gi >= nbins) return;
    const uint32_t sh = (gi & 15u) << 1;
    const uint32_t flag = use_hot ? (hot[gi >> 4] >> sh) & 3u : 0u;     // binned mode never thins samples
    const float mult = hot_mult(flag);
    const u64 cell = __builtin_nontemporal_load(atom + gi);
    __builtin_nontemporal_store(0ull, atom + gi);
    float yf, uf, vf, df;
    unpack_cell(cell, yf, uf, vf, df);
    float4 o = out[gi];
    const float m255 = mult * FL_INV255;
    o.w = fmaf(df, mult, o.w);
    o.x = fmaf(yf, m255, o.x);
    o.y = fmaf(uf, m255, o.y);
    o.z = fmaf(vf, m255, o.z);
    out[gi] = o;
    uint32_t nf = (uint32_t)(o.w > 128.0f) + (uint32_t)(o.w > 512.0f) + (uint32_t)(o.w > 2048.0f);
    uint32_t word = nf << sh;
    word |= __shfl_xor(word, 1);
    word |= __shfl_xor(word, 2);
    word |= __shfl_xor(word, 4);
    word |= __shfl_xor(word, 8);
    if ((gi & 15u) == 0) hot[gi >> 4] = word;
}

// ---- host-side launchers --------------------------------------------------------------------
static size_t iter_lds_bytes(int nw, int acc, uint32_t rounds, uint32_t nbins, uint32_t sub_log2)
{
    size_t nt = (size_t)nw * 64, b = 2 * 3 * nt * 4;
    if (acc == 1 || acc == 3) b += (size_t)rounds * nt * (acc == 3 ? 6 : 4) + (nw == 4 ? FL_CNT_SETS : FL_CNT_SETS_BIG) * (size_t)((nbins + 1 + 3) & ~3u) * 4 + 16
             + 128 * 4;      // chunk totals (only the all-waves scan uses them; the operand table sits behind them either way)
    else b += FL_PAL_W * 8;
    return b + ((size_t)FL_XTAB_BYTES << sub_log2);      // (sub-blocks: an operand table each)
}

void launch_iter(hipStream_t st, int nw, bool count, int acc, uint32_t nslots,
                 const int32_t *prog, const float *params, const u64 *palette, fl_mwc *rng,
                 float4 *points, const uint32_t *hot, u64 *atom, float *out4, u64 *counters,
                 uint32_t astride, uint32_t aheight, uint32_t round0, uint32_t nrounds, uint32_t fuse,
                 uint32_t tiles_x, uint32_t nbins, uint32_t rounds_per_batch, uint32_t nbatch_total,
                 uint32_t *log, uint32_t *dir, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t sub_log2)
{
    if (sub_log2 != 0u && (4 << sub_log2) != nw) abort();      // sub-blocks are four waves
    BinGeom bg = {tiles_x, nbins, rounds_per_batch, nbatch_total, sub_log2};
    const size_t lds = iter_lds_bytes(nw, acc, rounds_per_batch, nbins, sub_log2);
#define LAUNCH(NW, C, A) do { \
        static unsigned long long attr_done = 0; \
        ensure_max_dynamic_lds((const void *)k_iter<NW, C, A>, attr_done); \
        /* the timing events bracket the kernel itself (recorded by the dispatch packet), not the launch call */ \
        hipExtLaunchKernelGGL((k_iter<NW, C, A>), dim3(nslots), dim3(NW * 64), lds, st, ev_start, ev_stop, 0, prog, params, palette, \
        rng, points, hot, atom, out4, counters, astride, aheight, round0, nrounds, fuse, bg, log, dir); } while (0)
#define LAUNCH_NW(C, A) do { if (nw == 4) LAUNCH(4, C, A); else if (nw == 8) LAUNCH(8, C, A); else LAUNCH(16, C, A); } while (0)
    if (acc == 2) LAUNCH_NW(false, 2);
    else if (acc == 1) { if (count) LAUNCH_NW(true, 1); else LAUNCH_NW(false, 1); }
    else if (acc == 3) { if (count) LAUNCH_NW(true, 3); else LAUNCH_NW(false, 3); }
    else { if (count) LAUNCH_NW(true, 0); else LAUNCH_NW(false, 0); }
#undef LAUNCH_NW
#undef LAUNCH
}

void launch_iter_fn(hipStream_t st, hipFunction_t fn, int nw, int acc, uint32_t nslots,
                    const int32_t *prog, const float *params, const u64 *palette, fl_mwc *rng,
                    float4 *points, const uint32_t *hot, u64 *atom, float *out4, u64 *counters,
                    uint32_t astride, uint32_t aheight, uint32_t round0, uint32_t nrounds, uint32_t fuse,
                    uint32_t tiles_x, uint32_t nbins, uint32_t rounds_per_batch, uint32_t nbatch_total,
                    uint32_t *log, uint32_t *dir, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t sub_log2)
{
    if (sub_log2 != 0u && (4 << sub_log2) != nw) abort();      // sub-blocks are four waves
    BinGeom bg = {tiles_x, nbins, rounds_per_batch, nbatch_total, sub_log2};
    const size_t lds = iter_lds_bytes(nw, acc, rounds_per_batch, nbins, sub_log2);
    void *args[] = {&prog, &params, &palette, &rng, &points, &hot, &atom, &out4, &counters, &astride, &aheight,
                    &round0, &nrounds, &fuse, &bg, &log, &dir};
    // (no hipFuncSetAttribute here: that API takes a host function pointer, not a module function; module launches
    // accept up to the device's 160 KB of dynamic LDS as they are — the 137 KB workgroups of the 8K geometry run
    // through this path in tests/test_gpu_fullsize.py::test_cfg5_full_size)
    (void)hipExtModuleLaunchKernel(fn, nslots * (uint32_t)nw * 64, 1, 1, (uint32_t)nw * 64, 1, 1, lds, st, args, nullptr, ev_start, ev_stop, 0);
}

void launch_flush(hipStream_t st, u64 *atom, float4 *out, uint32_t *hot, uint32_t nbins, bool use_hot)
{
    hipLaunchKernelGGL(k_flush, dim3((nbins + 255) / 256), dim3(256), 0, st, atom, out, hot, nbins, use_hot ? 1 : 0);
}

// The clears at the head of a frame (cuburn/render.py:321-328: accumulator, packed cells, hot flags,
// counters, walker points := NaN) in ONE launch instead of five memsets: at 2 ms per frame five
// 6 us fills with their launch gaps were 2 % of the frame.  nbins is a multiple of 16.
__global__ void __launch_bounds__(256)
k_clear_frame(float4 *__restrict__ front, u64 *__restrict__ atom, uint32_t *__restrict__ hot, u64 *__restrict__ counters,
              float4 *__restrict__ points, uint32_t nbins, uint32_t npoints)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < nbins) {
        front[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        atom[i] = 0ull;
        if ((i & 15u) == 0u) hot[i >> 4] = 0u;
    }
    if (i < npoints) {
        const float nanf_ = __uint_as_float(0x7fc00000u);
        points[i] = make_float4(nanf_, nanf_, nanf_, nanf_);
    }
    if (i < 4u) counters[i] = 0ull;
}

void launch_clear_frame(hipStream_t st, float4 *front, u64 *atom, uint32_t *hot, u64 *counters, float4 *points,
                        uint32_t nbins, uint32_t npoints)
{
    const uint32_t n = nbins > npoints ? nbins : npoints;
    hipLaunchKernelGGL(k_clear_frame, dim3((n + 255) / 256), dim3(256), 0, st, front, atom, hot, counters, points, nbins, npoints);
}

void launch_shuffle_tap(hipStream_t st, int nw, uint32_t *out, uint32_t round)
{
    if (nw == 4) hipLaunchKernelGGL(k_shuffle_tap<4>, dim3(1), dim3(256), 0, st, out, round);
    else if (nw == 8) hipLaunchKernelGGL(k_shuffle_tap<8>, dim3(1), dim3(512), 0, st, out, round);
    else hipLaunchKernelGGL(k_shuffle_tap<16>, dim3(1), dim3(1024), 0, st, out, round);
}
#endif  // !FL_RTC
