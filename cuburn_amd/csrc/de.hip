// de.hip — one direction of the density-estimation filter as ONE kernel (gfx950).
//
// The reference runs, per direction, den_blur -> den_blur_1c -> bilateral through textures
// (cuburn/filters.py:62-95; cuburn/code/filters.py:106-131,166-264).  The first MI355X form
// (filters.hip: k_den_blur2_lds + k_de_bilateral_pk) kept that split and moved three per-pixel
// planes between the two kernels.  Here a direction reads the normalised image N = (x/w, y/w, z/w, w)
// once and writes it once; everything else lives in LDS for the lifetime of a tile:
//
//   * Sheared tiles.  Row j of a tile (and of its halo) starts S(j) = floor(j*K/2) pixels further
//     right, K/2 being the direction's x step per row, so that every tap of a pixel, every tap of
//     the two density blurs at a tap, and their neighbours land in (almost) the same COLUMN of the
//     staged parallelogram: the halo is 24 rows (13 for the directions that advance two pixels per
//     row) by 0..2 columns.  The horizontal direction uses 8 x 64 tiles with a 24-column halo.
//   * Both density blurs (7 taps at step 1, 7 taps at step 2) are evaluated in LDS for every staged
//     position a tap can reach, in the reference's summation order.
//   * The gradient factor exp2(-exp2(+-gspeed * (next.w - prev.w) / (avg + 1e-6))) of a tap
//     depends on the tap POSITION and the sign of r only (cuburn/code/filters.py:241-245) — for
//     the four directions with integer steps.  There the inner exponentials are two per-pixel
//     planes H+ / H- computed once per staged pixel, and the 31-tap loop keeps ONE v_exp_f32 per
//     tap.  For the half-slope directions next / prev depend on r mod 4 (the tap offsets are
//     rounded), so the inner exponential stays in the loop (two v_exp_f32 per tap).
//   * w^dpow is computed when a pixel is staged (2 transcendentals per staged pixel instead of a
//     4-byte plane read and written per direction).
//   * What bounds it (round 4): vector-ALU issue slots.  SQ_ACTIVE_INST_VALU of a direction equals its duration, the tap loop's own
//     instruction mix runs at 1.46 ns per wave-instruction per SIMD in isolation (tools/vgpr_bank_bench.hip) and the kernels reach ~80 %
//     of that; LDS conflicts, ILP, packed FMAs and workgroups per CU measured nothing, three instructions fewer per tap -6 %
//     (profiles/r04_de_instruction_experiments.txt, DESIGN.md 4.3 "Round 4").
//   * (round 3's reading, tools/valu_bench.hip wall-clock + tools/de_phases.py): a workgroup's life is a staging
//     phase that mostly waits (global loads, barriers: 2.6-4.6 us) and a tap phase that computes (the taps of all
//     resident workgroups together run the vector ALU at ~80 % of its measured peak of one wave instruction per
//     1.21 ns per SIMD); a kernel then pays ~8-10 us of ramp-up and tail on top (every workgroup stages at the start,
//     few are left at the end).  The tap arithmetic is arranged for the fewest instructions.  The colour
//     distance |n_q - c|^2 is expanded: cs*|n_q|^2 is a per-pixel plane S, -2*cs*c a per-centre
//     vector C', and cs*|c|^2 a per-centre scalar added to S (NOT factored out of the loop: for narrow
//     colour kernels the partial exponent overflows, see de_tap_loop):
//         e = S_q + cs*|c|^2 + n_q . C'   (1 add + 3 fma; "dead" tap or centre: a select to cs/2)
//           - | |ds|*pw_c - |ds|*pw_q |  - H(q)          (|ds|*w^dpow is a per-pixel plane too)
//     17 vector instructions per tap, one of them the exponential.
//   * Image edges.  The reference clamps every texture fetch (cuburn/code/filters.py:22-35), which
//     makes the blurred density at a tap outside the image the blur AT the clamped position, not
//     the blur of the clamped image.  Staged positions outside the image (they exist only in
//     border tiles) get their two blur values from a direct evaluation on the global image.
//
// Scalar (non-packed) math: on gfx950 v_pk_fma_f32 issues at ~1.6x the cost of v_fma_f32
// (tools/valu_bench.hip), which does not pay for the 16-apart pixel pairing and ds_read2_b32
// traffic the packed form needs; at <= 64 VGPRs a CU holds 32 waves (in workgroups of 256 threads, see DE_TW_).
#include "flame_device.h"
#include "kernels.h"
#include "tone_device.h"
#include <utility>
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstring>

// the image a direction reads / writes
struct DeImgPlain {
    float4 *p;
    __device__ __forceinline__ float4 ld(uint32_t i) const { return const_cast<const float4 *>(p)[i]; }
    __device__ __forceinline__ void st(uint32_t i, float4 v) const { p[i] = v; }
};
struct DeCoefs { float k[7]; float k2[19]; };      // the blur's 7 taps; both blurs as ONE 19-tap kernel (integer-step directions)
struct DeSpatial { float s[16]; };      // exp(-r^2 / (sqrt2 * sstd)), r = 0..15 (cuburn/code/filters.py:176-178), computed on the host

// ---- compile-time geometry of a direction --------------------------------------------------
// cuburn/code/filters.py:8-17,26-34: tap offset = round-to-nearest-even of slope * r
__host__ __device__ constexpr int de_num_x(int P) { constexpr int n[8] = {2, 0, 2, -2, 2, -1, 2, 1}; return n[P]; }
__host__ __device__ constexpr int de_num_y(int P) { constexpr int n[8] = {0, 2, 2, 2, 1, 2, -1, 2}; return n[P]; }
__host__ __device__ constexpr int de_rne_half(int v)      // v / 2 rounded to nearest, ties to even
{
    return (v % 2 == 0) ? v / 2 : ((v - 1) / 2 % 2 == 0 ? (v - 1) / 2 : (v + 1) / 2);
}
__host__ __device__ constexpr int de_dx(int P, int r) { return de_rne_half(de_num_x(P) * r); }
__host__ __device__ constexpr int de_dy(int P, int r) { return de_rne_half(de_num_y(P) * r); }
// shear: x shift of tile row j = floor(j * K / 2)
__host__ __device__ constexpr int de_k(int P) { constexpr int k[8] = {0, 0, 2, -2, 4, -1, -4, 1}; return k[P]; }
__host__ __device__ constexpr int de_floor_half(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }
__host__ __device__ constexpr int de_shear(int P, int j) { return de_floor_half(j * de_k(P)); }
// column displacement in sheared coordinates of an image displacement (dx, dy) from a row of parity par
__host__ __device__ constexpr int de_dv(int P, int par, int dx, int dy)
{
    return dx - (de_shear(P, par + dy) - de_shear(P, par));
}
__host__ __device__ constexpr bool de_hoisted(int P) { return P < 4; }      // integer steps: H+ / H- planes
// Tile shapes (output pixels = threads of a workgroup).  Round 2 used 32 x 32 (8 x 128 for the horizontal
// direction): two 1024-thread workgroups per CU.  A workgroup alternates a staging phase that mostly WAITS
// (global loads, four barriers) and a tap phase that computes, and a CU holds 32 waves whatever their grouping:
// smaller workgroups put more independent phases on a CU (512 threads: four per CU; 256: eight), and the
// kernels are 10-18 % faster (profiles/r03_de_tile_shapes.txt).  Taller tiles (less halo) are slower.
#ifndef DE_TW_
#define DE_TW_ 8       /* directions 4..7 (half slopes): 32 rows x 8 columns, 256 threads (round 3: 32 x 16, 512 threads) */
#endif
#ifndef DE_TH_
#define DE_TH_ 32
#endif
#ifndef DE_TWE_
#define DE_TWE_ DE_TW_ /* directions 4 and 6 (two pixels per row: even shear, consecutive rows per wave) */
#endif
#ifndef DE_TWH_
#define DE_TWH_ 8      /* directions 1..3 (integer steps, no column halo): 32 x 8, 256 threads */
#endif
#ifndef DE_THH_
#define DE_THH_ 32
#endif
#ifndef DE_TW0_
#define DE_TW0_ 64     /* the horizontal direction: DE_TH0_ x DE_TW0_ tiles (round 2: 8 x 128, round 3: 8 x 64) */
#endif
#ifndef DE_TH0_
#define DE_TH0_ 4
#endif
// Round 4: every direction in 256-thread workgroups, eight to a CU.  The DE alone is 3 % faster than with round 3's mix of
// 256 and 512 threads, the two-lane frame loop 4.7 % (1.458 -> 1.390 ms at cfg2: smaller workgroups and LDS blocks find room
// beside the other lane's kernels; profiles/r04_de_shapes_frame.txt).
#ifndef DE_MINW
#define DE_MINW 8      /* waves per SIMD the kernels are compiled for (8: 64 registers) */
#endif
#ifndef DE_PRIO_STAGE
#define DE_PRIO_STAGE 3
#endif
#ifndef DE_FAST_PREP
#define DE_FAST_PREP 1
#endif
#ifndef DE_X_NOBORDER_EVAL
#define DE_X_NOBORDER_EVAL 0  /* timing build (wrong results at the image edges): no direct evaluation of the blurs at clamped positions */
#endif
#ifndef DE_SKIP_OUTSIDE
#define DE_SKIP_OUTSIDE 1
#endif
#ifndef DE_BOTTOM_FIRST
#define DE_BOTTOM_FIRST 1
#endif
#ifndef DE_RUN
#define DE_RUN 8u             /* tile order 2: tiles per run (a run stays on one XCD) */
#endif
#ifndef DE_INTERIOR_LOADS
#define DE_INTERIOR_LOADS 1   /* tiles that touch no image edge load their staged region without clamps, one address per thread */
#endif
#ifndef DE_LANE_PAIRS
#define DE_LANE_PAIRS 1       /* 8-pixel rows of the half-slope directions: two rows 8 slots apart per hardware lane group */
#endif
struct DeReach { int hu, hv; };
// Largest row / column displacement (sheared coordinates) of any staged value a tile pixel needs:
// tap r, then the second blur's tap 2i there, then the first blur's tap j there (each offset is
// rounded on its own, as the reference's nested clamped fetches are).  blur = false: taps only.
__host__ __device__ constexpr DeReach de_reach(int P, bool blur)
{
    int hu = 0, hv = 0;
    for (int par = 0; par < 2; ++par)
        for (int r = -16; r <= 16; ++r)
            for (int i = -3; i <= 3; ++i)
                for (int j = -3; j <= 3; ++j) {
                    if (!blur && (i != 0 || j != 0 || r == 16 || r == -16)) continue;
                    if ((r == 16 || r == -16) && (i != 0 || j != 0 || de_hoisted(P))) continue;   // +-16: next / prev density only
                    int u = de_dy(P, r), v = de_dv(P, par, de_dx(P, r), de_dy(P, r));
                    int p1 = (par + u) & 1;
                    v += de_dv(P, p1, de_dx(P, 2 * i), de_dy(P, 2 * i)); u += de_dy(P, 2 * i);
                    int p2 = (par + u) & 1;
                    v += de_dv(P, p2, de_dx(P, j), de_dy(P, j)); u += de_dy(P, j);
                    hu = (u < 0 ? -u : u) > hu ? (u < 0 ? -u : u) : hu;
                    hv = (v < 0 ? -v : v) > hv ? (v < 0 ? -v : v) : hv;
                }
    return DeReach{hu, hv};
}

template <int P> struct DeGeo {
    static constexpr int K = de_k(P);
    // output tile
    static constexpr int TW = P == 0 ? DE_TW0_ : (de_hoisted(P) ? DE_TWH_ : (P == 4 || P == 6) ? DE_TWE_ : DE_TW_);
    static constexpr int TH = P == 0 ? DE_TH0_ : (de_hoisted(P) ? DE_THH_ : DE_TH_);
    static constexpr int NT = TW * TH;                  // threads of a workgroup: one output pixel each
    static_assert(NT % 64 == 0 && NT <= 1024 && (P == 0 ? TW % 64 == 0 : 64 % TW == 0 && TH % (128 / TW) == 0), "whole waves, rows of equal parity per wave");
    static constexpr bool HOIST = de_hoisted(P);
    // staged region (densities): everything a tile pixel's taps and their blurs can reach
    static constexpr int HU = de_reach(P, true).hu, HV = de_reach(P, true).hv;
    static constexpr int ROWS = TH + 2 * HU, COLS = TW + 2 * HV, NPX = ROWS * COLS;
    // Staging loops give every thread ONE column of the staged region and every RS-th row (RS whole rows per
    // iteration, the threads beyond RS * COLS idle): element = it * RS * COLS + tid, row = it * RS + tid / COLS — the
    // division happens once per thread.  (With element = it * NT + tid and COLS = 18 or 20, every element of every loop
    // paid its own division by a constant: 17 % of all vector instructions of the half-slope directions were integer.)
    // (The horizontal direction keeps element = it * NT + tid: two iterations of 112-pixel rows, divisions that cost
    // little, and a border path that has no registers left for per-thread constants.)
    static constexpr bool ROWWISE = P != 0;
    // (odd K: an even number of rows per iteration, so that a thread's rows keep their parity — its tap offsets — and its
    // global address advances by the same amount from iteration to iteration: floor((u + RS) * K / 2) = floor(u * K / 2) + RS * K / 2)
    static constexpr int RS = (K & 1) ? (NT / COLS) & ~1 : NT / COLS, NACT = ROWWISE ? RS * COLS : NT;
    static constexpr int NIT = ROWWISE ? (ROWS + RS - 1) / RS : (NPX + NT - 1) / NT;
    // plane A (normalised pixels) holds only the rows the taps themselves read (r = -16 .. 16), with the staged
    // region's columns: the outer rows are needed as densities for the blurs, not as pixels
    static constexpr int HA = de_dy(P, 16) < 0 ? -de_dy(P, 16) : de_dy(P, 16);
    static constexpr int AROWS = TH + 2 * HA, NPXA = AROWS * COLS, AOFF = (HU - HA) * COLS;      // A element = staged element - AOFF
    static_assert(HA <= HU, "the taps reach no further than the blurs");
    // plane B (per-pixel tap terms): the positions of the taps themselves
    static constexpr int HBU = de_reach(P, false).hu, HBV = de_reach(P, false).hv;
    // BSTR: plane B's row stride.  A ds_read_b128 is served in four groups of 16 lanes, each conflict-free when its 16 float4 slots
    // differ mod 16; with 8-pixel rows a group holds two rows (de_out_px), which must then lie 8 slots apart mod 16: rows two apart
    // on a stride of 12 (plane A of directions 4 / 6: 8 + 2 * 2 columns) or of 10 with the equal-parity rows of directions 5 / 7 — plane B
    // of directions 4 / 6 has 10 columns and rows two apart, hence two columns of padding (round 5: its reads took 12 LDS cycles, not 4).
    static constexpr int BROWS = TH + 2 * HBU, BCOLS = TW + 2 * HBV;
    static constexpr int BPAD = (!de_hoisted(P) && (K & 1) == 0 && TW == 8 && DE_LANE_PAIRS) ? 12 - BCOLS : 0;
    static constexpr int BSTR = BCOLS + BPAD, NPXB = BROWS * BSTR;
    static_assert(BPAD >= 0 && (BPAD == 0 || (2 * BSTR) % 16 == 8), "rows two apart must lie 8 float4 slots apart mod 16");
    static constexpr int RSB = (K & 1) ? (NT / BCOLS) & ~1 : NT / BCOLS, NACTB = ROWWISE ? RSB * BCOLS : NT;
    static constexpr int NITB = ROWWISE ? (BROWS + RSB - 1) / RSB : (NPXB + NT - 1) / NT;
    // LDS: A float4[NPXA] | B float4[NPXB] | (integer-step directions) the fast path's density plane float[NPX];
    // the nested preparation's two dense float planes live in B's space
    static constexpr int MINW = DE_MINW;
    static constexpr size_t LDS = (size_t)(NPXA + NPXB) * 16 + (de_hoisted(P) ? (size_t)NPX * 4 : 0) + 64;
    static constexpr int SPAN = de_shear(P, TH - 1) < 0 ? -de_shear(P, TH - 1) : de_shear(P, TH - 1);
    // Frame tables (tiles that touch an image edge, see "frame tables" in the kernel): every staged position outside the image clamps
    // to a position ON the image's frame — (0 | xmax, clamped row) or (x, 0 | ymax) — so a tile needs the blurs at no more than
    // 2 * ROWS + 2 * XS frame positions: one per staged row and side, one per x of the staged region's extent and side.
    static constexpr int SHMIN = de_shear(P, -HU) < de_shear(P, ROWS - 1 - HU) ? de_shear(P, -HU) : de_shear(P, ROWS - 1 - HU);
    static constexpr int SHMAX = de_shear(P, -HU) < de_shear(P, ROWS - 1 - HU) ? de_shear(P, ROWS - 1 - HU) : de_shear(P, -HU);
    static constexpr int XS = COLS + SHMAX - SHMIN, NF = 2 * ROWS + 2 * XS;
    static_assert((2 * NPX + 2 * NF) * 4 <= NPXB * 16, "preparation planes + frame tables must fit into plane B's space");
    // element offset of image displacement (dx, dy) from a position in a row of parity par
    static constexpr int off(int par, int dx, int dy) { return dy * COLS + de_dv(P, par, dx, dy); }
    static constexpr int offb(int par, int dx, int dy) { return dy * BSTR + de_dv(P, par, dx, dy); }
    static constexpr int tap_off(int par, int r) { return off(par, de_dx(P, r), de_dy(P, r)); }
    static constexpr int tap_offb(int par, int r) { return offb(par, de_dx(P, r), de_dy(P, r)); }
    static constexpr int min_tap_off(int par)
    {
        int m = 0;
        for (int r = -16; r <= 16; ++r) m = tap_off(par, r) < m ? tap_off(par, r) : m;
        return m;
    }
    static constexpr int min_tap_offb(int par)
    {
        int m = 0;
        for (int r = -15; r <= 15; ++r) m = tap_offb(par, r) < m ? tap_offb(par, r) : m;
        return m;
    }
};

__device__ __forceinline__ int de_clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// Direct evaluation of the two density blurs at an in-image position (border tiles only)
template <int P, class IMG>
__device__ float de_b1_global(const IMG &N, const fl_dim &d, int cx, int cy, const DeCoefs &k)
{
    float den = 0.0f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int x = de_clampi(cx + de_dx(P, j - 3), 0, (int)d.astride - 1), y = de_clampi(cy + de_dy(P, j - 3), 0, (int)d.ah - 1);
        den += N.ld((uint32_t)(y * (int)d.astride + x)).w * k.k[j];
    }
    return den;
}
template <int P, class IMG>
__device__ float de_b2_global(const IMG &N, const fl_dim &d, int cx, int cy, const DeCoefs &k)
{
    // all 49 loads are issued before the first is waited for: as a loop of seven dependent rounds this was seven
    // memory round trips (5-7 us) for the one tile that needs it, with the rest of its workgroup at the barrier
    float den = 0.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int x = de_clampi(cx + de_dx(P, 2 * (i - 3)), 0, (int)d.astride - 1), y = de_clampi(cy + de_dy(P, 2 * (i - 3)), 0, (int)d.ah - 1);
        den += de_b1_global<P>(N, d, x, y, k) * k.k[i];
    }
    return den;
}
// ... both blurs at once: the second blur's middle term (i = 3, t(0) = 0) IS the first blur at the position itself
template <int P, class IMG>
__device__ float de_b12_global(const IMG &N, const fl_dim &d, int cx, int cy, const DeCoefs &k, float &b1)
{
    float den = 0.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int x = de_clampi(cx + de_dx(P, 2 * (i - 3)), 0, (int)d.astride - 1), y = de_clampi(cy + de_dy(P, 2 * (i - 3)), 0, (int)d.ah - 1);
        const float v = de_b1_global<P>(N, d, x, y, k);
        if (i == 3) b1 = v;
        den += v * k.k[i];
    }
    return den;
}

// The 31-tap loop of one output pixel.  PAR = parity of the pixel's tile row (matters for odd K).
//
// Left to itself the compiler either hoists all LDS reads of the unrolled loop or sinks the
// five accumulation chains below them (700 bytes of scratch per lane either way), so the reads
// are explicit ds_read asm with immediate offsets from one base address per plane, in a
// two-taps-per-step, double-buffered software pipeline: step g issues the reads of step g+1,
// then waits (s_waitcnt lgkmcnt(n), n = the reads just issued) for its own.  The waiting asm
// takes the step's buffer AND the accumulators as read-write operands: that is what pins the
// arithmetic of step g-1 before it and the arithmetic of step g after it.
// thread -> output pixel of its tile: a wave covers two rows of equal parity (row parity selects the tap offsets when K
// is odd); the horizontal direction: 64 consecutive pixels of one row
template <int P>
__device__ __forceinline__ void de_out_px(int wv, int lane, int &ou, int &ov)
{
    using G = DeGeo<P>;
    if (P == 0) { constexpr int WPR = G::TW / 64; ou = wv / WPR; ov = (wv % WPR) * 64 + lane; }
    else if (G::TW == 8 && !G::HOIST && DE_LANE_PAIRS) {
        // 8-pixel rows, eight to a wave (the half-slope directions).  A hardware group of 16 lanes (quads {0, 3, 5, 6} or {1, 2, 4, 7}
        // of each half of the wave) takes TWO rows, two apart among the wave's eight: on the planes' strides (DeGeo::BSTR) their
        // float4 slots are 8 apart mod 16 and every read of the tap loop is served in 4 LDS cycles (tools/de_geometry_model.py;
        // before: A 8 / B 12 cycles for directions 4 / 6, 8 / 8 for 5 / 7).
        const int q = (lane & 31) >> 2, grp = (q ^ (q >> 1) ^ (q >> 2)) & 1, rank = q >> 1;
        const int gi = 2 * (lane >> 5) + grp;                                 // group within the wave
        const int k = (gi >> 1) * 4 + (gi & 1) + 2 * (rank >> 1);             // row among the wave's eight
        ov = ((rank & 1) << 2) | (lane & 3);
        ou = (G::K & 1) ? (wv >> 1) * 16 + (wv & 1) + 2 * k : wv * 8 + k;
    }
    else if (G::K & 1) {
        constexpr int RPW = 64 / G::TW;                                   // rows per wave (equal parity)
        ou = (wv >> 1) * (2 * RPW) + (wv & 1) + 2 * (lane / G::TW); ov = lane % G::TW;
    } else {
        // even K: no parity to respect, so a wave takes CONSECUTIVE rows — with 8-pixel rows (128 bytes) two rows two
        // apart start on the same LDS bank, and the 16 lanes a ds_read_b128 serves together span two rows
        // (SQ_LDS_BANK_CONFLICT was 40 % of the LDS cycles of direction 1 with the equal-parity mapping)
        ou = wv * (64 / G::TW) + lane / G::TW; ov = lane % G::TW;
    }
}
// the lane number from an instruction the compiler cannot move or share between the arms of a branch
__device__ __forceinline__ int de_lane_here()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
// A at tap r+1 (the next pixel), B at tap r.  Plane B holds, per staged pixel q (Kp = cs/2, the exponent of a "dead" pair):
//   x = |ds| * w_q^dpow,  y = 1 if w_q > 0 else 0,
//   z, w = cs*|n_q|^2 + Kp - H+(q),  cs*|n_q|^2 + Kp - H-(q)      (integer-step directions: the hoisted gradient terms)
//   z, w = cs*|n_q|^2 + Kp,          gspeed / (avg(q) + 1e-6)     (half-slope directions: the gradient term stays in the loop)
struct DeTap { f4v a, b; };

#ifdef DE_X_NOLDS      /* timing build: the tap loop without its LDS reads (results are garbage) */
#define DE_RD128(dst, addr, boff) asm volatile("; no read %1 %2" : "=v"(dst) : "v"(addr), "n"(boff) : "memory")
#else
#define DE_RD128(dst, addr, boff) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(boff) : "memory")
#endif

template <int P> __host__ __device__ constexpr int de_tap_reads(int r)       // LDS reads issued for tap r
{
    if (r > 15) return 0;
    return ((r < 15 || !de_hoisted(P)) ? 1 : 0) + (r != 0 ? 1 : 0);          // A of tap r+1; B of tap r (the centre's own terms are in registers)

}

// SLOW = false: every centre of the wave is live (w_c > 0) — all but the rim of a flame.  Then the exponent of a pair is
//     e = B.zw(q) + y_q * (cs*|c|^2 - Kp) + n_q . C'  - | |ds|pw_c - |ds|pw_q |        (1 + 3 fma, 2 sub: 6 instructions)
// (a dead tap has n_q = 0, y_q = 0 and cs*|n_q|^2 = 0: Kp - H remains, the reference's colour difference of 0.5), against
// add + 3 fma + compare + select + 2 sub + sub H = 9 before the gradient term and the liveness moved into the plane:
// 13 vector instructions + 1 exponential per tap (integer-step directions), 16 + 2 (half-slope directions).
// SLOW = true: a wave with a dead centre anywhere.  A dead centre pairs with EVERY tap at colour difference 0.5, so the
// tap's own cs*|n_q|^2 has to leave the exponent again (4 more instructions; C' and the centre term are 0 for such a lane).
template <int P, int PAR, bool SLOW>
__device__ __forceinline__ void de_tap_loop(const float4 *__restrict__ sA, const float4 *__restrict__ sB, int wv,
                                            float cs2, const DeSpatial &spk, float4 &res)
{
    using G = DeGeo<P>;
    // The forms of the loop sit in the arms of wave-uniform branches.  Each arm finds its pixel itself, from the wave
    // number (a scalar) and a lane number the compiler cannot share between arms: otherwise it computes the arms' common
    // per-centre prologue once, ahead of the branches, and spills what it cannot hold for all of them.
    int ou, ov;
    de_out_px<P>(wv, de_lane_here(), ou, ov);
    const int ci = (ou + G::HA) * G::COLS + ov + G::HV;
    const int cb = (ou + G::HBU) * G::BSTR + ov + G::HBV;
    constexpr int MINOFF = G::min_tap_off(PAR), MINOFFB = G::min_tap_offb(PAR);
#define TOFF(r) (G::tap_off(PAR, (r)) - MINOFF)
#define TOFFB(r) (G::tap_offb(PAR, (r)) - MINOFFB)
    static_assert((G::tap_off(PAR, 16) - MINOFF) * 16 < 65536 && (G::tap_off(PAR, -16) - MINOFF) >= 0, "tap offsets must fit the DS immediate");
    const float4 *__restrict__ bA = sA + (ci + MINOFF);
    const float4 *__restrict__ bB = sB + (cb + MINOFFB);
    // LDS byte addresses of the two planes' bases (dynamic LDS starts at the kernel's LDS base)
    uint32_t aA = (uint32_t)(size_t)bA, aB = (uint32_t)(size_t)bB;
    const float4 cen = bA[TOFF(0)];
    // the reference normalises the centre with 1/(w + 1e-6) and the taps with 1/w
    const float cfix = cen.w * frcp(cen.w + 1.0e-6f);
    const float ccx = cen.x * cfix, ccy = cen.y * cfix, ccz = cen.z * cfix;
    const bool cen_live = cen.w > 0.0f;
    // cs*|n_q - c|^2 = cs*|n_q|^2 + n_q . C' + cs*|c|^2.  The last term is the same for every tap of this
    // centre, but it cannot be factored out of the loop: for a narrow colour kernel (cstd < ~0.03) or
    // colours above 1 the partial exponent reaches +150 and 2^(cs*|c|^2) underflows — the sums overflow
    // to inf and come back as NaN / 0 (found by tools/soak_filters.py; the full exponent is never positive).
    // (-2 cs is wave-uniform and the same in every form of the loop: left alone the compiler computes it once, ahead of the forms'
    // branches, and holds — in one kernel spills — a vector register across all of them)
    float cs2h = cs2;
    asm volatile("" : "+s"(cs2h));
    const float m2u = -2.0f * cs2h;
    const float m2 = cen_live ? m2u : 0.0f;
    float Cx = ccx * m2, Cy = ccy * m2, Cz = ccz * m2;
    const float Kp = 0.5f * cs2;
    float Dl = cen_live ? cs2 * fmaf(ccz, ccz, fmaf(ccy, ccy, ccx * ccx)) - Kp : 0.0f;     // y_q * Dl: the centre term of a live pair
    float dcs = cen_live ? 0.0f : cs2;                                                       // SLOW: removes cs*|n_q|^2 for a dead centre
    // the centre's own pair (r = 0): its plane terms from registers (no gradient term, no density difference)
    const float y0c = fmaf(cs2, fmaf(cen.z, cen.z, fmaf(cen.y, cen.y, cen.x * cen.x)), Kp) + Dl;
    float cds = bB[TOFFB(0)].x;                               // |ds| * w_c^dpow
    float wprev = G::HOIST ? 0.0f : bA[TOFF(-16)].w;
    const float4 p0 = bA[TOFF(-15)];
    f4v pix = {p0.x, p0.y, p0.z, p0.w};
    f2v oxy = {0.0f, 0.0f}, ozw = {0.0f, 0.0f};               // (sum f*w*nx, sum f*w*ny), (sum f*w*nz, sum f*w)
    float wsum = 0.0f;

    // (the rare form is not pipelined: one buffer, a step's reads issued and waited for in the step itself — sixteen
    // registers less, which is what lets both forms live in one kernel of 64 registers without scratch)
    constexpr int NBUF = SLOW ? 1 : 2;
    DeTap L[NBUF][2];
    auto issue = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
#define ISSUE_TAP(k) if constexpr (-15 + g * 2 + (k) <= 15) { \
            constexpr int r = -15 + g * 2 + (k); \
            if constexpr (r < 15 || !G::HOIST) DE_RD128(L[g % NBUF][k].a, aA, TOFF(r + 1) * 16); \
            if constexpr (r != 0) DE_RD128(L[g % NBUF][k].b, aB, TOFFB(r) * 16); }
        ISSUE_TAP(0) ISSUE_TAP(1)
#undef ISSUE_TAP
    };
    auto step = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        if constexpr (SLOW) issue(gc);
        else if constexpr (g + 1 < 16) issue(std::integral_constant<int, g + 1>{});
        constexpr int inflight = (!SLOW && g + 1 < 16) ? de_tap_reads<P>(-15 + (g + 1) * 2) + de_tap_reads<P>(-14 + (g + 1) * 2) : 0;
        DeTap (&T)[2] = L[g % NBUF];
#define DE_WAIT_OPS "+v"(T[0].a), "+v"(T[0].b), "+v"(T[1].a), "+v"(T[1].b), "+v"(aA), "+v"(aB), "+v"(oxy), "+v"(ozw), "+v"(wsum), \
                    "+v"(pix), "+v"(wprev), "+v"(Cx), "+v"(Cy), "+v"(Cz), "+v"(Dl), "+v"(cds)
        if constexpr (SLOW) asm volatile("s_waitcnt lgkmcnt(%[n])" : DE_WAIT_OPS, "+v"(dcs) : [n] "n"(inflight));
        else asm volatile("s_waitcnt lgkmcnt(%[n])" : DE_WAIT_OPS : [n] "n"(inflight));
#undef DE_WAIT_OPS
        // The two taps of a step go through the arithmetic in LOCKSTEP, stage by stage: a tap is a chain of ~10 dependent
        // instructions, and a wave that issues dependent instructions back to back gets one issue slot per ~10 clocks
        // (tools/valu_bench.hip: 1.56 ns per wave-instruction per SIMD for a dependent chain against 1.05-1.23 for
        // independent ones, even with eight waves per SIMD).  Left alone the compiler schedules tap after tap.
        constexpr int r0 = -15 + g * 2, r1 = r0 + 1;
        constexpr bool two = r1 <= 15;                                   // the last step has one tap
        const f4v px0 = pix, px1 = T[0].a;                               // tap r's pixel came with tap r-1's read
        f4v b0 = T[0].b, b1 = T[1].b;
        float t0, t1 = 0.0f;
        // colour term: plane value + the centre term of a live pair + n_q . C'
        if constexpr (r0 == 0) t0 = SLOW ? fmaf(cs2, fmaf(px0.z, px0.z, fmaf(px0.y, px0.y, px0.x * px0.x)), 0.5f * cs2) + Dl : y0c;
        else t0 = fmaf(b0.y, Dl, G::HOIST ? (r0 < 0 ? b0.w : b0.z) : b0.z);
        if constexpr (two) {
            if constexpr (r1 == 0) t1 = SLOW ? fmaf(cs2, fmaf(px1.z, px1.z, fmaf(px1.y, px1.y, px1.x * px1.x)), 0.5f * cs2) + Dl : y0c;
            else t1 = fmaf(b1.y, Dl, G::HOIST ? (r1 < 0 ? b1.w : b1.z) : b1.z);
        }
        t0 = fmaf(px0.z, Cz, t0); if constexpr (two) t1 = fmaf(px1.z, Cz, t1);
        t0 = fmaf(px0.y, Cy, t0); if constexpr (two) t1 = fmaf(px1.y, Cy, t1);
        t0 = fmaf(px0.x, Cx, t0); if constexpr (two) t1 = fmaf(px1.x, Cx, t1);
        if constexpr (SLOW) {                                            // a dead centre: the tap's own cs*|n_q|^2 leaves again
            if constexpr (r0 != 0) t0 = fmaf(fmaf(px0.z, px0.z, fmaf(px0.y, px0.y, px0.x * px0.x)), -dcs, t0);
            if constexpr (two && r1 != 0) t1 = fmaf(fmaf(px1.z, px1.z, fmaf(px1.y, px1.y, px1.x * px1.x)), -dcs, t1);
        }
        // density term (none for the centre's own pair)
        float e0 = t0, e1 = t1;
        if constexpr (r0 != 0) { const float d0 = cds - b0.x; e0 = t0 - fabsf(d0); }
        if constexpr (two && r1 != 0) { const float d1 = cds - b1.x; e1 = t1 - fabsf(d1); }
        // gradient term of the half-slope directions: next.w - prev.w around the tap, b.w = gspeed / (avg + 1e-6)
        if constexpr (!G::HOIST) {
            float g0 = 0.0f, g1 = 0.0f;
            if constexpr (r0 != 0) g0 = (T[0].a.w - wprev) * b0.w;
            if constexpr (two && r1 != 0) g1 = (T[1].a.w - px0.w) * b1.w;
            if constexpr (r0 != 0) g0 = fexp2(r0 < 0 ? -g0 : g0);
            if constexpr (two && r1 != 0) g1 = fexp2(r1 < 0 ? -g1 : g1);
            if constexpr (r0 != 0) e0 -= g0;
            if constexpr (two && r1 != 0) e1 -= g1;
        }
        float f0 = fexp2(e0), f1 = two ? fexp2(e1) : 0.0f;
        f0 *= spk.s[r0 < 0 ? -r0 : r0]; if constexpr (two) f1 *= spk.s[r1 < 0 ? -r1 : r1];
        const float fw0 = f0 * px0.w, fw1 = f1 * px1.w;
        wsum += f0;
        oxy.x = fmaf(fw0, px0.x, oxy.x); oxy.y = fmaf(fw0, px0.y, oxy.y); ozw.x = fmaf(fw0, px0.z, ozw.x); ozw.y += fw0;
        if constexpr (two) {
            wsum += f1;
            oxy.x = fmaf(fw1, px1.x, oxy.x); oxy.y = fmaf(fw1, px1.y, oxy.y); ozw.x = fmaf(fw1, px1.z, ozw.x); ozw.y += fw1;
            wprev = px1.w;
            if (r1 < 15 || !G::HOIST) pix = T[1].a;
        }
    };
    if constexpr (!SLOW) issue(std::integral_constant<int, 0>{});
    [&]<int... Gs>(std::integer_sequence<int, Gs...>) __attribute__((always_inline)) {
        (step(std::integral_constant<int, Gs>{}), ...);
    }(std::make_integer_sequence<int, 16>{});
#undef TOFF
#undef TOFFB
    // out.xyz = sum f*w*n, out.w = sum f*w: the normalised colour is their ratio (the 1/weightsum of the
    // reference cancels), the density is out.w / (weightsum + 1e-10)
    const float ow = ozw.y;
    const float wn = ow * frcp(wsum + 1e-10f);
    const float rn = ow >= 1.17549435e-38f ? frcp(ow) : 0.0f;   // v_rcp_f32 of a denormal is +inf
    res = make_float4(oxy.x * rn, oxy.y * rn, ozw.x * rn, wn);
}

#ifndef DE_SLOW_INLINE
#define DE_SLOW_INLINE __forceinline__
#endif
// The rare form
template <int P, int PAR>
__device__ DE_SLOW_INLINE void de_tap_loop_slow(const float4 *__restrict__ sA, const float4 *__restrict__ sB, int wv,
                                              float cs2, const DeSpatial &spk, float4 &res)
{
    de_tap_loop<P, PAR, true>(sA, sB, wv, cs2, spk, res);
}

// Normalise the accumulator into N (first pass input); with YUV -> RGB in front when the chain starts with `yuv`
__device__ __forceinline__ float4 de_yuv_px(float4 p)       // cuburn/code/filters.py:71-77 + cuburn/code/color.py:25-40
{
    const float u = p.y - 0.5f * p.w, v = p.z - 0.5f * p.w;
    return make_float4(fmaxf(0.0f, p.x + 1.402f * v), fmaxf(0.0f, p.x - 0.34414f * u - 0.71414f * v), fmaxf(0.0f, p.x + 1.772f * u), p.w);
}
// IN: what the first direction finds in `N`: 0 = the normalised image (x/w, y/w, z/w, w); 1 = the raw
// accumulator, normalised as it is staged; 2 = the raw accumulator in YUV (the chain starts with `yuv`).
// OUT: 1 = the last direction un-normalises its result and applies the tone filters that follow the DE
// in the chain (DeTail) as it stores it.  Both run the per-pixel device functions of the separate
// kernels (k_de_normalise, k_de_finish_tone) in the same order: bit-identical results, two passes over
// the image and two launches less.
template <int IN>
__device__ __forceinline__ float4 de_in_px(float4 p)
{
    if (IN == 0) return p;
    if (IN == 2) p = de_yuv_px(p);
    const float rw = p.w > 0.0f ? frcp(p.w) : 0.0f;
    return make_float4(p.x * rw, p.y * rw, p.z * rw, p.w);
}

// -DDE_X_PHASES: every workgroup adds the 100 MHz ticks it spent in each phase to de_phase_ticks[direction][phase]
// (0 = until the loads are in LDS, 1 = first blur, 2 = tap terms, 3 = plane B written, 4 = taps + store, 5 = workgroups);
// fl_debug_de_phases reads and clears them (tools/de_phases.py).
#if defined(DE_X_PHASES)
#define DE_PH_MAXWG 16384
__device__ unsigned long long de_phase_rec[8][DE_PH_MAXWG][6];      // per workgroup: no two writers share a word
#define DE_PHASE(n) do { if (threadIdx.x == 0 && blockIdx.x < DE_PH_MAXWG) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); de_phase_rec[P][blockIdx.x][n] += now_ - tick_; tick_ = now_; } } while (0)
extern "C" __attribute__((visibility("default"))) int fl_debug_de_phases(unsigned long long *out, int clear)
{
    static unsigned long long host[8][DE_PH_MAXWG][6];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(de_phase_rec), sizeof host) != hipSuccess) return -1;
    for (int p = 0; p < 8; ++p) for (int n = 0; n < 6; ++n) { unsigned long long a = 0; for (int w = 0; w < DE_PH_MAXWG; ++w) a += host[p][w][n]; out[p * 6 + n] = a; }
    if (clear) { if (hipMemset(nullptr, 0, 0) != hipSuccess) (void)hipGetLastError(); void *sym = nullptr; if (hipGetSymbolAddress(&sym, HIP_SYMBOL(de_phase_rec)) != hipSuccess || hipMemset(sym, 0, sizeof host) != hipSuccess) return -1; }
    return 0;
}
#else
#define DE_PHASE(n)
#endif

#define DE_IN_PX(p) de_in_px<IN>(p)
template <int P, int IN, int OUT>
__global__ void __launch_bounds__(DeGeo<P>::NT, DeGeo<P>::MINW)      // 8 waves per SIMD (<= 64 registers): 32 waves per CU in workgroups of NT threads
k_de_dir(fl_dim d, float4 *__restrict__ Nout_, const float4 *__restrict__ N_, DeCoefs kc, DeSpatial spk,
         float cs2, float ads, float dpow, float gspeed, uint32_t tiles_y, uint32_t ntiles, uint32_t tiles_x, uint32_t magic, DeTail tail)
{
    using G = DeGeo<P>;
    static_assert(2 * G::NPX * 4 <= G::NPXB * 16, "the preparation planes must fit into plane B's space");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *sA = reinterpret_cast<float4 *>(smem);
    float4 *sB = reinterpret_cast<float4 *>(smem + (size_t)G::NPXA * 16);
    float *sW = reinterpret_cast<float *>(sB);                   // nested prep: dense density plane ...
    float *s1 = sW + G::NPX;                                     // ... and first blur (both die before B is written)
    float *sF1 = s1 + G::NPX, *sF2 = sF1 + G::NF;                // frame tables (border tiles): first / second blur at frame positions
    float *sWf = reinterpret_cast<float *>(smem + (size_t)(G::NPXA + G::NPXB) * 16);      // fast prep: density plane beside B
    const DeImgPlain N = {const_cast<float4 *>(N_)}, Nout = {Nout_};

    // XCD-aware tile order: workgroup b runs on XCD b % 8; give every XCD a contiguous run of
    // tiles in column-major order, so that the tiles resident together on an XCD are vertical
    // neighbours and find each other's halo rows in that XCD's L2.
    // Tiles at the image's sides are the slow ones (staged positions outside the image evaluate their blurs on
    // the global image): XCDs 4..7 walk their run backwards, so that the right edge is done first, not last.
    // (one division per workgroup, by tiles_y or tiles_x, as a multiplication by the host's `magic` = floor(2^32 / divisor) + 1: the
    // three 32-bit divisions this used to take were ~100 dependent scalar instructions in front of a workgroup's first load)
    const uint32_t per_xcd = (ntiles + 7u) / 8u, xcd = blockIdx.x & 7u;
    int tx, ty;
    if (tail.order == 0) {
        const uint32_t t = xcd * per_xcd + (xcd < 4u ? (blockIdx.x >> 3) : per_xcd - 1u - (blockIdx.x >> 3));
        if (t >= ntiles) return;
        const uint32_t q = tiles_y == 1u ? t : __umulhi(t, magic);      // t / tiles_y (a divisor of 1 has no 32-bit magic)
        tx = (int)q; ty = (int)(t - q * tiles_y);
    } else {
        // tail.order = 1 (FLAME_DE_ORDER=1): plain row-major order — tile t = workgroup t, rows of tiles left to right.
        // tail.order = 2: row-major in RUNS — DE_RUN consecutive tiles of a row of tiles go to ONE XCD, the next run to the next XCD.
        // Plain row-major order deals neighbouring tiles to eight different L2s, and the flat, wide staged regions of directions
        // 0 / 4 / 6 share most of their cache lines with their left and right neighbours: 4.2x the image fetched into the L2s per
        // launch, real HBM traffic once the image has outgrown the Infinity Cache.
        const uint32_t loc = blockIdx.x >> 3;
        const uint32_t rm = tail.order == 1 ? blockIdx.x : ((loc / DE_RUN) * 8u + xcd) * DE_RUN + loc % DE_RUN;
        if (rm >= ntiles) return;
        const uint32_t q = tiles_x == 1u ? rm : __umulhi(rm, magic);    // rm / tiles_x
        ty = (int)q; tx = (int)(rm - q * tiles_x);
        // (the bottom row of tiles first: its tiles touch the image's edge — the slow ones, frame tables — and dispatched last they
        // WERE the kernel's tail; the top row, the other slow one, has always been first)
        if (DE_BOTTOM_FIRST) ty = ty == 0 ? (int)tiles_y - 1 : ty - 1;
    }
    // x of column 0 of tile row 0; for K > 0 the band starts SPAN to the left so that its last row reaches x = 0
    const int bx0 = tx * G::TW - (G::K > 0 ? G::SPAN : 0), by0 = ty * G::TH;
    const int tid = threadIdx.x;
    const int xmax = (int)d.astride - 1, ymax = (int)d.ah - 1;
    // does any staged position leave the image?  (block-uniform)
    const bool border = by0 - G::HU < 0 || by0 + G::TH + G::HU > (int)d.ah ||
                        bx0 + min(0, de_shear(P, -G::HU)) + min(0, de_shear(P, G::TH + G::HU)) - G::HV - 1 < 0 ||
                        bx0 + max(0, de_shear(P, -G::HU)) + max(0, de_shear(P, G::TH + G::HU)) + G::TW + G::HV + 1 > (int)d.astride;

    // this thread's row (within an iteration's RS rows) and column of the staged region / of plane B: the only divisions
#define DE_S_THREAD() int ts_ = tid; asm volatile("" : "+v"(ts_)); \
    const int sr0 = G::ROWWISE ? ts_ / G::COLS : 0, sc0 = ts_ - sr0 * G::COLS; const bool sact = ts_ < G::NACT; \
    /* element `it` of this thread: index in the staged region, row, column; false when there is none */ \
    auto s_elem = [&](int it, int &idx, int &ul, int &vl) __attribute__((always_inline)) -> bool { \
        idx = it * G::NACT + tid; \
        if (G::ROWWISE) { ul = it * G::RS + sr0; vl = sc0; return sact && ul < G::ROWS; } \
        ul = idx / G::COLS; vl = idx - ul * G::COLS; return idx < G::NPX; }; \
    /* the same, clamped to the region's last element (loads are issued by every thread) */ \
    auto s_elem_clamped = [&](int it, int &ul, int &vl) __attribute__((always_inline)) { \
        if (G::ROWWISE) { ul = min(it * G::RS + sr0, G::ROWS - 1); vl = sc0; } \
        else { const int idx = min(it * G::NT + tid, G::NPX - 1); ul = idx / G::COLS; vl = idx - ul * G::COLS; } }
    // (plane B's are found again by every phase that needs them, from a thread number the compiler cannot connect
    // with the earlier ones: held from the top of the kernel they cost the border tiles' blur evaluation its registers)
#define DE_B_THREAD() int tb_ = tid; asm volatile("" : "+v"(tb_)); \
    const int br0 = G::ROWWISE ? tb_ / G::BCOLS : 0, bc0 = tb_ - br0 * G::BCOLS; const bool bact = tb_ < G::NACTB; \
    const int bidx0 = (br0 + G::HU - G::HBU) * G::COLS + bc0 + G::HV - G::HBV;      /* staged element of this thread's plane-B element, iteration 0 */ \
    const int bsto0 = br0 * G::BSTR + bc0;                                          /* ... and where it is stored (rows of BSTR float4) */ \
    /* plane-B element `it` of this thread: its index in the plane, its staged element, staged row and column */ \
    auto b_elem = [&](int it, int &bidx, int &idx, int &ul, int &vl) __attribute__((always_inline)) -> bool { \
        if (G::ROWWISE) { const int ub = it * G::RSB + br0; ul = ub + G::HU - G::HBU; vl = bc0 + G::HV - G::HBV; \
                          bidx = bsto0 + it * G::RSB * G::BSTR; \
                          idx = bidx0 + it * G::RSB * G::COLS; return bact && ub < G::BROWS; } \
        static_assert(G::ROWWISE || G::BSTR == G::BCOLS, "padding only with row-wise staging"); \
        bidx = it * G::NACTB + tid; \
        const int ub = bidx / G::BCOLS, vb = bidx - ub * G::BCOLS; \
        ul = ub + G::HU - G::HBU; vl = vb + G::HV - G::HBV; idx = ul * G::COLS + vl; return bidx < G::NPXB; }

    // Timing builds (results are garbage): -DDE_X_TAPSONLY runs the taps on whatever the LDS holds, -DDE_X_STOP_AFTER=n
    // ends the workgroup after staging phase n (1..4) — tools/ab_de.sh, profiles/r03_de_phases.txt.
    // Staging is a few instructions between long waits (global loads, barriers), the taps are 550 instructions
    // back to back: with the arbiter's default order a young workgroup's staging instructions queue behind the
    // older workgroups' tap loops and its loads go out late.  Staging therefore runs at raised priority, the
    // taps at the default: the loads of the next tiles are in flight while the current tiles compute.
    __builtin_amdgcn_s_setprio(DE_PRIO_STAGE);
#if defined(DE_X_PHASES)
    unsigned long long tick_ = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x < DE_PH_MAXWG) de_phase_rec[P][blockIdx.x][5] += 1ull;
#endif
#ifdef DE_X_TAPSONLY
    if (gspeed != 12345.0f) goto taps;
#endif
#ifdef DE_X_STOP_AFTER
#define DE_X_STOP(n) if (DE_X_STOP_AFTER == n && gspeed != 12345.0f) { if (tid == 0 && sB[17].x == 1.2345e-33f) Nout.st(0, sB[17]); return; }
#else
#define DE_X_STOP(n)
#endif
    // ---- integer-step directions, no staged position outside the image: two phases instead of four -----------
    // The density blurs collapse into one 19-tap kernel on the staged densities (kc.k2), so the first blur's plane,
    // its phase and its barrier go, and plane B can be written as it is computed (nothing aliases it): load ->
    // barrier -> tap terms -> barrier.  The regrouped sum differs from the nested one by float rounding only (1e-7
    // relative in `avg`); tiles that touch an image edge keep the nested form below, which follows the reference's
    // clamped fetches literally.
    if (G::HOIST && !border && DE_FAST_PREP) {
        float4 tq[G::NIT];
        DE_S_THREAD();
#pragma unroll
        for (int it = 0; it < G::NIT; ++it) {
            int ul, vl;
            s_elem_clamped(it, ul, vl);
            const int gx = bx0 + (((ul - G::HU) * G::K) >> 1) + vl - G::HV, gy = by0 + ul - G::HU;      // inside the image: no clamps
            tq[it] = DE_IN_PX(N.ld((uint32_t)(gy * (int)d.astride + gx)));
        }
#pragma unroll
        for (int it = 0; it < G::NIT; ++it) {
            int idx, ul, vl;
            if (s_elem(it, idx, ul, vl)) {
                sWf[idx] = tq[it].w;
                if (idx >= G::AOFF && idx < G::AOFF + G::NPXA) sA[idx - G::AOFF] = tq[it];
            }
        }
        __syncthreads();
        DE_X_STOP(1)
        DE_PHASE(0);
        DE_B_THREAD();
#pragma unroll
        for (int it = 0; it < G::NITB; ++it) {
            int bidx, idx, ul, vl;
            if (!b_elem(it, bidx, idx, ul, vl)) continue;
            float den = 0.0f;
#pragma unroll
            for (int m = -9; m <= 9; ++m) den = fmaf(sWf[idx + G::off(0, de_dx(P, m), de_dy(P, m))], kc.k2[m + 9], den);
            const float ra = frcp(den + 1.0e-6f) * gspeed;
            const float4 n = sA[idx - G::AOFF];
            constexpr int dn = G::off(0, de_dx(P, 1), de_dy(P, 1));
            const float g = (sWf[idx + dn] - sWf[idx - dn]) * ra;
            const float yk = fmaf(cs2, fmaf(n.z, n.z, fmaf(n.y, n.y, n.x * n.x)), 0.5f * cs2);
            sB[bidx] = make_float4(ads * de_pow(n.w, dpow), n.w > 0.0f ? 1.0f : 0.0f, yk - fexp2(g), yk - fexp2(-g));
        }
        __syncthreads();
        DE_X_STOP(4)
        DE_PHASE(2);
    } else {
    // ---- S0: stage N (edge-clamped) and the dense density plane ------------------------------
    float4 tn[G::NIT];
    {
    DE_S_THREAD();
    if (G::ROWWISE && !border && DE_INTERIOR_LOADS) {
        // no staged position leaves the image: no clamps, and a thread's element of iteration `it` lies it * RS rows below its
        // first one — one address per thread, a wave-uniform step per iteration (RS * K is even, see DeGeo::RS)
        const int u0 = sr0 - G::HU;
        const uint32_t g0 = (uint32_t)((by0 + u0) * (int)d.astride + bx0 + ((u0 * G::K) >> 1) + sc0 - G::HV);
        const uint32_t gstep = (uint32_t)(G::RS * (int)d.astride + G::RS * G::K / 2);
#pragma unroll
        for (int it = 0; it < G::NIT; ++it) {
            const int nrows = G::ROWS - it * G::RS < G::RS ? G::ROWS - it * G::RS : G::RS;       // rows of this iteration (threads beyond them idle)
            if (sr0 < nrows) tn[it] = DE_IN_PX(N.ld(g0 + (uint32_t)it * gstep));
            else tn[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    } else {
#pragma unroll
    for (int it = 0; it < G::NIT; ++it) {
        int ul, vl;
        s_elem_clamped(it, ul, vl);
        const int gx = de_clampi(bx0 + (((ul - G::HU) * G::K) >> 1) + vl - G::HV, 0, xmax);
        const int gy = de_clampi(by0 + ul - G::HU, 0, ymax);
        tn[it] = DE_IN_PX(N.ld((uint32_t)(gy * (int)d.astride + gx)));
    }
    }
    // ---- frame tables --------------------------------------------------------------------------
    // A staged position outside the image takes the blurs AT its clamped position (the reference clamps the fetch from the blurred
    // texture), which has to be evaluated on the global image: 7 loads for the first blur, 49 for both.  Round 4 did that wherever a
    // staged position was virtual — every wave-iteration of S1 / S2 with one such lane ran the whole sequence, and at 1080p, where one
    // tile in six touches an edge, this was 52 of the chain's 414 us (profiles/r05_de_border.txt).  All positions outside the image
    // clamp to positions ON its frame, many to the same one: the tile evaluates each frame position it can need ONCE (one thread per
    // position, its loads issued behind the tile's own, which are still in flight) and S1 / S2 look the values up.
    // Same functions on the same arguments as before: same bits.
    if (border && !DE_X_NOBORDER_EVAL) {
        const int x0a = bx0 - G::HV, nrows_top = min(G::ROWS, G::HU - by0), first_bot = max(0, ymax + 1 - (by0 - G::HU));
        // x extent of the staged rows above / below the image (shear is monotonic in the row)
        int tlo = 1, thi = 0, blo = 1, bhi = 0;
        if (nrows_top > 0) {
            const int a = x0a + de_shear(P, -G::HU), b = x0a + (((nrows_top - 1 - G::HU) * G::K) >> 1);
            tlo = min(a, b); thi = max(a, b) + G::COLS - 1;
        }
        if (first_bot < G::ROWS) {
            const int a = x0a + (((first_bot - G::HU) * G::K) >> 1), b = x0a + de_shear(P, G::ROWS - 1 - G::HU);
            blo = min(a, b); bhi = max(a, b) + G::COLS - 1;
        }
        for (int e = tid; e < G::NF; e += G::NT) {
            int fx, fy; bool need;
            if (e < 2 * G::ROWS) {                                       // (0 | xmax, row): rows whose staged span sticks out on that side
                const int side = e >= G::ROWS, u = e - side * G::ROWS;
                const int x0 = x0a + (((u - G::HU) * G::K) >> 1);
                need = side ? x0 + G::COLS - 1 > xmax : x0 < 0;
                fx = side ? xmax : 0; fy = de_clampi(by0 + u - G::HU, 0, ymax);
            } else {                                                     // (x, 0 | ymax): x inside the image, under a staged row above / below it
                const int e2 = e - 2 * G::ROWS, side = e2 >= G::XS;
                fx = x0a + G::SHMIN + e2 - side * G::XS; fy = side ? ymax : 0;
                need = fx >= 0 && fx <= xmax && (side ? (fx >= blo && fx <= bhi) : (fx >= tlo && fx <= thi));
            }
            if (need) {
                float b1;
                sF2[e] = de_b12_global<P>(N, d, fx, fy, kc, b1);
                sF1[e] = b1;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < G::NIT; ++it) {
        int idx, ul, vl;
        if (s_elem(it, idx, ul, vl)) {
            sW[idx] = tn[it].w;
            if (idx >= G::AOFF && idx < G::AOFF + G::NPXA) sA[idx - G::AOFF] = tn[it];
        }
    }
    }
    __syncthreads();
    DE_X_STOP(1)
    DE_PHASE(0);

    // ---- S1: first density blur (7 taps, step 1) ----------------------------------------------
    // Every staged position is evaluated; where a tap leaves the staged region it reads whatever
    // lies next to the plane inside this workgroup's LDS (plane A below, the blur plane above) and
    // the value is meaningless — by construction of the halo (de_reach) no tile pixel ever needs it.
    {
    DE_S_THREAD();
#pragma unroll
    for (int it = 0; it < G::NIT; ++it) {
        int idx, ul, vl;
        if (!s_elem(it, idx, ul, vl)) continue;
        const bool par = ((ul - G::HU) & 1) != 0;
        float den = 0.0f;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int o0 = G::off(0, de_dx(P, j - 3), de_dy(P, j - 3)), o1 = G::off(1, de_dx(P, j - 3), de_dy(P, j - 3));
            const int o = (G::K & 1) ? (par ? o1 : o0) : o0;
            den = fmaf(sW[idx + o], kc.k[j], den);
        }
        if (border && !DE_X_NOBORDER_EVAL) {
            const int gxu = bx0 + (((ul - G::HU) * G::K) >> 1) + vl - G::HV, gyu = by0 + ul - G::HU;
            if (gxu < 0 || gxu > xmax || gyu < 0 || gyu > ymax)         // virtual position: the blur AT the clamped position
                den = sF1[gxu < 0 ? ul : gxu > xmax ? G::ROWS + ul : 2 * G::ROWS + (gyu < 0 ? 0 : G::XS) + gxu - (bx0 - G::HV + G::SHMIN)];
        }
        s1[idx] = den;
    }
    }
    __syncthreads();
    DE_X_STOP(2)
    DE_PHASE(1);

    // ---- S2: per-pixel tap terms for every position a tap can land on -------------------------
    // second blur (7 taps, step 2) -> gspeed / (avg + 1e-6) -> gradient exponentials; |ds| * w^dpow;
    // cs * |n|^2.  Held in registers until every thread is done with the preparation planes.
    float4 pb[G::NITB];
    {
    DE_B_THREAD();
#pragma unroll
    for (int it = 0; it < G::NITB; ++it) {
        pb[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int bidx, idx, ul, vl;
        if (!b_elem(it, bidx, idx, ul, vl)) continue;
        const bool par = ((ul - G::HU) & 1) != 0;
        float den = 0.0f;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int o0 = G::off(0, de_dx(P, 2 * (i - 3)), de_dy(P, 2 * (i - 3))), o1 = G::off(1, de_dx(P, 2 * (i - 3)), de_dy(P, 2 * (i - 3)));
            const int o = (G::K & 1) ? (par ? o1 : o0) : o0;
            den = fmaf(s1[idx + o], kc.k[i], den);
        }
        if (border && !DE_X_NOBORDER_EVAL) {
            const int gxu = bx0 + (((ul - G::HU) * G::K) >> 1) + vl - G::HV, gyu = by0 + ul - G::HU;
            if (gxu < 0 || gxu > xmax || gyu < 0 || gyu > ymax)
                den = sF2[gxu < 0 ? ul : gxu > xmax ? G::ROWS + ul : 2 * G::ROWS + (gyu < 0 ? 0 : G::XS) + gxu - (bx0 - G::HV + G::SHMIN)];
        }
        const float ra = frcp(den + 1.0e-6f) * gspeed;
        const float4 n = sA[idx - G::AOFF];
        pb[it].x = ads * de_pow(n.w, dpow);
        pb[it].y = n.w > 0.0f ? 1.0f : 0.0f;
        const float yk = fmaf(cs2, fmaf(n.z, n.z, fmaf(n.y, n.y, n.x * n.x)), 0.5f * cs2);
        if (G::HOIST) {
            // next / prev of a tap at this position are its neighbours one step along the direction
            // (integer steps: t(r+1) - t(r) = t(1) for every r); staged values are edge-clamped, so
            // this is next.w - prev.w of the reference for virtual positions as well
            constexpr int dn = G::off(0, de_dx(P, 1), de_dy(P, 1));
            const float g = (sW[idx + dn] - sW[idx - dn]) * ra;
            pb[it].z = yk - fexp2(g);
            pb[it].w = yk - fexp2(-g);
        } else {
            pb[it].z = yk;
            pb[it].w = ra;
        }
    }
    }
    __syncthreads();
    DE_X_STOP(3)
    DE_PHASE(2);
    // ---- S3: the per-pixel plane replaces the preparation planes -------------------------------
    {
    DE_B_THREAD();
#pragma unroll
    for (int it = 0; it < G::NITB; ++it) {
        int bidx, idx, ul, vl;
        if (b_elem(it, bidx, idx, ul, vl)) sB[bidx] = pb[it];
    }
    }
    __syncthreads();
    DE_X_STOP(4)
    DE_PHASE(3);

    }
#ifdef DE_X_TAPSONLY
taps:
#endif
    __builtin_amdgcn_s_setprio(0);
    // ---- taps --------------------------------------------------------------------------------
    // (the wave number travels through the tap loop in a scalar register and the lane number is recomputed after it:
    // no vector register is live across the loop — its forms leave none to spare)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    do {                                                            // (a block to `continue` out of)
    float4 res;
    {
        int cu, cv;
        de_out_px<P>(wv, tid & 63, cu, cv);
        // The parallelogram sticks out of the image at both ends of a band (and the last row of tiles below it): a wave none of whose
        // outputs lies inside has helped to stage the tile and is done — 4.5 % of the waves at 1080p (round 5).
        {
            const int xq = bx0 + ((cu * G::K) >> 1) + cv, yq = by0 + cu;
            if (DE_SKIP_OUTSIDE && __builtin_amdgcn_ballot_w64(xq >= 0 && xq <= xmax && yq <= ymax) == 0ull) continue;
        }
        // a wave with a dead centre (w_c = 0: the rim of the flame, sparse images) takes the form that handles one
        const bool slow = __builtin_amdgcn_ballot_w64(!(sA[(cu + G::HA) * G::COLS + cv + G::HV].w > 0.0f)) != 0ull;
        if ((G::K & 1) && (wv & 1)) { if (slow) de_tap_loop_slow<P, 1>(sA, sB, wv, cs2, spk, res); else de_tap_loop<P, 1, false>(sA, sB, wv, cs2, spk, res); }
        else { if (slow) de_tap_loop_slow<P, 0>(sA, sB, wv, cs2, spk, res); else de_tap_loop<P, 0, false>(sA, sB, wv, cs2, spk, res); }
    }
    int ou, ov;
    de_out_px<P>(wv, de_lane_here(), ou, ov);

    const int xo = bx0 + ((ou * G::K) >> 1) + ov, yo = by0 + ou;
    if (xo >= 0 && xo <= xmax && yo <= ymax) {                      // the parallelogram sticks out of the image at both ends of a band
        if (OUT) {                                                  // as k_de_finish_tone (filters.hip)
            float4 p = make_float4(res.x * res.w, res.y * res.w, res.z * res.w, res.w);
            if (tail.do_log) p = logscale_px(p, tail.k1, tail.k2);
            if (tail.do_clip) p = colorclip_px(p, tail.vib, tail.highpow, tail.gam, tail.lin, tail.lingam);
            res = p;
        }
        Nout.st((uint32_t)(yo * (int)d.astride + xo), res);
    }
    } while (false);
    DE_PHASE(4);
}

template <int P, int IN, int OUT>
static void launch_de_dir_one(hipStream_t st, fl_dim d, float4 *Nout, const float4 *N, DeCoefs kc, DeSpatial spk,
                              float cs2, float ads, float dpow, float gspeed, DeTail tail)
{
    using G = DeGeo<P>;
    static_assert(G::LDS * (G::MINW * 256 / G::NT) <= 160 * 1024, "LDS must allow MINW waves per SIMD");
    static unsigned long long attr = 0;
    ensure_max_dynamic_lds((const void *)k_de_dir<P, IN, OUT>, attr);
    const uint32_t tiles_x = (d.astride + G::SPAN + G::TW - 1) / G::TW, tiles_y = (d.ah + G::TH - 1) / G::TH;
    const uint32_t ntiles = tiles_x * tiles_y;
    const uint32_t gran = tail.order == 2 ? 8u * DE_RUN : 8u;      // whole runs on every XCD
    const uint32_t divisor = tail.order == 0 ? tiles_y : tiles_x, magic = (uint32_t)(0x100000000ull / divisor) + 1u;
    if ((unsigned long long)(ntiles + gran) * divisor >= 0x100000000ull) abort();      // __umulhi(n, magic) is n / divisor for n * divisor < 2^32
    hipLaunchKernelGGL((k_de_dir<P, IN, OUT>), dim3(gran * ((ntiles + gran - 1) / gran)), dim3(G::NT), G::LDS, st, d, Nout, N, kc, spk,
                       cs2, ads, dpow, gspeed, tiles_y, ntiles, tiles_x, magic, tail);
}

// in_mode (pattern 0 only; ignored by the others): 1 = N is the raw accumulator, 2 = the raw YUV accumulator (yuv -> rgb first) — the first
// direction always normalises as it stages; anything else is refused (a "0 = already normalised" form existed until round 5: a caller
// that still relied on it would get a second normalisation).  tail (pattern 7 only, may be null): un-normalise + the tone filters riding along.
void launch_de_dir(hipStream_t st, fl_dim d, int pattern, float4 *Nout, const float4 *N, const float *coefs7,
                   float sstd, float cstd, float dstd, float dpow, float gspeed, int in_mode, const DeTail *tail)
{
    DeCoefs kc;
    for (int i = 0; i < 7; ++i) kc.k[i] = coefs7[i];
    // blur2(blur1(w))(q) = sum_i k_i sum_j k_j w(q + t(2i) + t(j)) = sum_m K_m w(q + t(m)), K_m = sum_{2i + j = m} k_i k_j,
    // wherever t(a) + t(b) = t(a + b): the four directions with integer steps, away from the image edges
    for (int m = -9; m <= 9; ++m) {
        float a = 0.0f;
        for (int i = -3; i <= 3; ++i) { const int j = m - 2 * i; if (j >= -3 && j <= 3) a += coefs7[i + 3] * coefs7[j + 3]; }
        kc.k2[m + 9] = a;
    }
    // per-launch scalars (cuburn/code/filters.py:176-183), evaluated once on the host
    DeSpatial spk;
    for (int r = 0; r < 16; ++r) spk.s[r] = expf((float)(r * r) / (-1.41421353816986f * sstd));
    const float cs2 = 1.0f / (-1.41421353816986f * 3.0f * cstd) * 1.44269502162933f;      // exp(c*x) = exp2(c*log2e*x)
    const float ads = fabsf(-0.5f / dstd);
    DeTail none = {};
    // Tile order, per direction and image size (profiles/r05_de_order.txt; round 4's table: profiles/r04_de_order_prof.txt).
    //   0: per-XCD column-major runs (vertical neighbours, which share halo rows, on one XCD's L2) — directions 1, 5, 7, whose
    //      staged regions are tall and narrow;
    //   2: row-major in runs of DE_RUN tiles per XCD — directions 0, 4, 6 (flat, wide staged regions) and the diagonals 2, 3
    //      (rows of a tile start one pixel further along: horizontal neighbours share most cache lines).  Round 4 ran 0 / 4 / 6 in
    //      plain row-major order (1), which deals neighbouring tiles to eight different L2s: 4.2x the image fetched per launch.
    // Above the Infinity Cache's 256 MiB (8K: 537 MB per image) a line fetched twice is HBM traffic twice, and directions 5 / 7 are
    // better off in runs as well (8K: 854-864 us against 867-875; 4K and 1080p: 0-3 % the other way).
    // 1080p: 415 us per chain with this table against 430 with round 4's; 4K 1428 / 1442; 8K 5699 / 5807.
    // FLAME_DE_ORDER: one digit for every direction, or eight digits, one per direction.
    static const char *forced_s = getenv("FLAME_DE_ORDER");
    const int forced = !forced_s || !*forced_s ? -1 : (strlen(forced_s) == 8 ? forced_s[pattern] - '0' : forced_s[0] - '0');
    const bool beyond_mall = (size_t)d.astride * d.ah * 16u > ((size_t)256 << 20);
    const int table = (pattern == 1) ? 0 : (pattern == 5 || pattern == 7) ? (beyond_mall ? 2 : 0) : 2;
    const int order = forced >= 0 && forced <= 2 ? forced : table;
    none.order = order;
    DeTail tl = tail ? *tail : none;
    tl.order = order;
#define DE(P, I, O) launch_de_dir_one<P, I, O>(st, d, Nout, N, kc, spk, cs2, ads, dpow, gspeed, tl)
    switch (pattern) {
    case 0:                                                                   // the first direction normalises the accumulator (after yuv -> rgb, if asked)
        if (in_mode == 2) DE(0, 2, 0); else if (in_mode == 1) DE(0, 1, 0); else abort();
        break;
    case 1: DE(1, 0, 0); break;
    case 2: DE(2, 0, 0); break;
    case 3: DE(3, 0, 0); break;
    case 4: DE(4, 0, 0); break;
    case 5: DE(5, 0, 0); break;
    case 6: DE(6, 0, 0); break;
    case 7: DE(7, 0, 1); break;                                                // the last one un-normalises (+ the tone filters of `tail`)
    default: break;
    }
#undef DE
}

