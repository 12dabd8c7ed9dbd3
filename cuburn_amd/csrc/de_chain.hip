// de_chain.hip — all eight directions of the density-estimation filter in ONE persistent launch (gfx950).
//
// The reference runs the directions one after the other (cuburn/filters.py:62-95); de.hip does the same with one
// kernel per direction.  A direction's kernel at 1080p is ~4.2 "rounds" of workgroups (8540 tiles over 2048
// resident workgroups): its last round runs a fifth full and its first one starts with every workgroup staging at
// once — eight times per chain.  Direction t+1 of a tile only needs direction t of the tiles within its staged reach
// (24 rows, a few dozen columns), so here the tiles of all eight directions form one ordered work list per XCD and the
// chip never drains between directions:
//
//   * 256-thread workgroups for every direction (tile shapes below), 8 per CU, each pulling the next item of its XCD's
//     list with one atomic (the list: direction 0's tiles of the XCD's stripe of the image, band by band, then
//     direction 1's, ...).  Pulling IN ORDER from lists in which every item depends only on items of the previous
//     direction makes the scheme deadlock-free: the earliest unfinished item of a list is always held by a running
//     workgroup whose own dependencies are earlier still.
//   * done[direction][band][stripe] counts finished tiles.  A tile waits (one wave polls, relaxed agent-scope loads,
//     s_sleep) until the counters of the previous direction's bands and stripes within its reach — the reach of ITS
//     staged region (read after write) and of THEIR staged regions over its output (write after read: the two images
//     ping-pong) — have reached their totals.
//   * Hand-off across CUs / XCDs (per-XCD L2s are not coherent, a CU's L1 is never refreshed by other CUs' stores):
//     tiles store write-through (sc1) and load with sc1 (L1 bypass); every storing wave drains its stores
//     (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, one lane bumps the counter with an agent-scope atomic
//     (cdna_hip_programming.md Guideline 16, R1).
//
// The same tile shapes are also available one direction per launch (k_de_one) so that the persistent launch can be
// compared bit for bit with eight launches of the same arithmetic.
//
// Measured (profiles/r04_de_chain.txt, r04_de_lap.txt): 950-970 us against 440-460 for eight plain launches.  Most of the loss is
// the list heads: 8540 RETURNING agent-scope atomics per head queue up at ~55 ns each, so a workgroup waits ~12 us for every item
// (a static assignment — workgroup w takes items w, w + 256, ... — runs the chain in 547 us, but two such launches from the two
// lanes, each partly resident, wait for each other's missing workgroups: the frame loop stalls until the waits give up; removed
// again).  The overlapped launches at the end of this file get the same overlap between directions from the hardware's own
// in-order dispatch.
#define DE_CHAIN_BUILD 1
// (the tile shapes of this file are its own, whatever an experiment build of de.hip is given on the command line)
#undef DE_TW_
#undef DE_TWE_
#undef DE_TH_
#undef DE_TWH_
#undef DE_THH_
#undef DE_TW0_
#undef DE_TH0_
#undef DE_LANE_GROUPS
#undef DE_OPT0_
#undef DE_OPTH_
#undef DE_OPT_
#undef DE_MINW
#undef DE_HPLANES
#define DE_HPLANES 0
#define DE_OPT0_ 1
#define DE_OPTH_ 1
#define DE_OPT_ 1
#define DE_MINW 8
#define DE_TW_ 8        /* directions 4..7: 32 x 8 */
#define DE_TWE_ 8
#define DE_TH_ 32
#define DE_TWH_ 8       /* directions 1..3: 32 x 8 */
#define DE_THH_ 32
#define DE_TW0_ 64      /* the horizontal direction: 4 x 64 */
#define DE_TH0_ 4
#define DE_LANE_GROUPS 0
#include "de.hip"

namespace {

constexpr int CH_NT = 256;
template <int P> constexpr size_t ch_lds() { return DeGeo<P>::LDS; }
constexpr size_t ch_max(size_t a, size_t b) { return a > b ? a : b; }
constexpr size_t CH_LDS = ch_max(ch_max(ch_max(ch_lds<0>(), ch_lds<1>()), ch_max(ch_lds<2>(), ch_lds<3>())),
                                 ch_max(ch_max(ch_lds<4>(), ch_lds<5>()), ch_max(ch_lds<6>(), ch_lds<7>())));
static_assert(DeGeo<0>::NT == CH_NT && DeGeo<1>::NT == CH_NT && DeGeo<4>::NT == CH_NT && DeGeo<5>::NT == CH_NT, "every direction in 256 threads");
static_assert((CH_LDS + 64) * 8 <= 160 * 1024, "eight workgroups per CU");

// geometry of a direction as run-time numbers (the dependency arithmetic mixes two directions)
struct ChDir { int tw, th, hu, reach_x, span; };
template <int P> constexpr ChDir ch_dir()
{
    using G = DeGeo<P>;
    // half-width of everything a tile stages beyond its own columns [tx * TW, tx * TW + TW): the band's lean (SPAN), the
    // shear over the halo rows, the column halo, slack
    return ChDir{G::TW, G::TH, G::HU, G::SPAN + (G::HU * (G::K < 0 ? -G::K : G::K) + 1) / 2 + G::HV + 3, G::SPAN};
}
__constant__ const ChDir CH_DIRS[8] = {ch_dir<0>(), ch_dir<1>(), ch_dir<2>(), ch_dir<3>(), ch_dir<4>(), ch_dir<5>(), ch_dir<6>(), ch_dir<7>()};
static const ChDir CH_DIRS_H[8] = {ch_dir<0>(), ch_dir<1>(), ch_dir<2>(), ch_dir<3>(), ch_dir<4>(), ch_dir<5>(), ch_dir<6>(), ch_dir<7>()};

// the filter's scalars live in device memory, not in the kernel's arguments: read through a pointer that is opaque in
// every iteration of the persistent loop they are loaded where a tile uses them — as kernel arguments all ~60 of them are
// loaded ahead of the loop and held across eight tile bodies (scalar registers spilt into vector lanes, those to scratch)
struct ChParams {
    DeCoefs kc; DeSpatial spk;
    float cs2, ads, dpow, gspeed;
    DeTail tail;
    int in_mode, has_tail;
    fl_dim d;
    float4 *img[2];                 // direction p reads img[p & 1], writes img[(p + 1) & 1]
    uint32_t nstripes;              // 8 (one list per XCD) or 1 (small images: one list)
    uint32_t maxbands;              // stride of the counters' band index
    uint32_t *heads;                // [8] next item of each list
    uint32_t *done;                 // [8][maxbands][8] finished tiles per direction, band, stripe
    uint32_t *fail;                 // set when a wait gives up (a bug, never expected)
    uint32_t tiles_x[8], tiles_y[8];
    uint32_t x0[8][9];              // first tile column of stripe s of direction p (x0[p][nstripes] = tiles_x[p])
    // overlapped launches (k_de_lap): nothing is cleared between chains — a tile is finished when its flag holds the chain's
    // epoch, a direction has been dispatched completely when its (ever-growing) count of started workgroups says so
    uint32_t epoch, maxtiles;
    uint32_t dbg;                   // FLAME_DE_LAP_DBG, timing experiments (wrong results): 1 no waits, 2 no publishing, 4 one stream and no gates, 8 no started count, 16 plain loads / stores
    uint32_t *flags;                // [8][maxtiles] epoch of the last chain in which tile t of direction p finished
    unsigned long long *started;    // [8][CH_NCNT] (one 64-byte line each) workgroups of direction p started since the scratch was allocated
};
__global__ void k_ch_params(ChParams *dst, ChParams v) { if (threadIdx.x == 0) *dst = v; }

// the parameter block is read through the scalar path (address space 4: s_load), like kernel arguments
typedef const __attribute__((address_space(4))) ChParams *ChP;
// every word workgroups share is accessed as GLOBAL memory with agent scope (never through a flat address)
typedef __attribute__((address_space(1))) uint32_t *ChG;
template <int P, class IMG>
__device__ __forceinline__ void ch_run(ChP q, const IMG &src, const IMG &dst, int tx, int ty, int tid)
{
    // (member-wise copies through the scalar path: every member is used with compile-time indices only, so the copies
    // become scalar registers loaded where they are used)
    fl_dim d; d.w = q->d.w; d.h = q->d.h; d.aw = q->d.aw; d.ah = q->d.ah; d.astride = q->d.astride;
    DeCoefs kc;
#pragma unroll
    for (int i = 0; i < 7; ++i) kc.k[i] = q->kc.k[i];
#pragma unroll
    for (int i = 0; i < 19; ++i) kc.k2[i] = q->kc.k2[i];
    DeSpatial spk;
#pragma unroll
    for (int i = 0; i < 16; ++i) spk.s[i] = q->spk.s[i];
    DeTail tail;
    tail.do_log = q->tail.do_log; tail.k1 = q->tail.k1; tail.k2 = q->tail.k2; tail.do_clip = q->tail.do_clip; tail.vib = q->tail.vib;
    tail.highpow = q->tail.highpow; tail.gam = q->tail.gam; tail.lin = q->tail.lin; tail.lingam = q->tail.lingam;
    de_tile<P, IMG>(d, dst, src, kc, spk, q->cs2, q->ads, q->dpow, q->gspeed, tail, P == 0 ? q->in_mode : 0,
                    P == 7 && q->has_tail != 0, tx, ty, tid);
}

// What lives ACROSS a tile body decides whether the eight bodies fit the 64 vector / 80 scalar registers that eight
// waves per SIMD allow, so it is kept to: the parameter block's address, the list number and the wave number (scalar).
// Everything else is re-read through an address the compiler cannot see through (asm), the thread number is rebuilt
// from the wave number and v_mbcnt, the buffer descriptors are built where they are used.
#ifdef CH_X_TIMES      /* per-workgroup 100 MHz ticks: [0] fetching items, [1] waiting for neighbours, [2] tiles, [3] publishing, [4] tiles done */
__device__ unsigned long long ch_times[2048][5];
#define CH_T(n) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); ch_times[blockIdx.x][n] += now_ - tick_; tick_ = now_; } } while (0)
#else
#define CH_T(n)
#endif
__global__ void __launch_bounds__(CH_NT, 8) k_de_chain(const ChParams *prm_)
{
#ifdef CH_X_TIMES
    unsigned long long tick_ = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the item's coordinates travel from lane 0 to the workgroup through the last words of the dynamic LDS (no static
    // LDS: it would shift the dynamic base off its 16-byte alignment, cdna_hip_programming.md Guideline 17)
    volatile int *bc = reinterpret_cast<volatile int *>(smem + CH_LDS);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t xcd = blockIdx.x & 7u;
    for (;;) {
        const ChParams *prm0 = prm_;
        asm volatile("" : "+s"(prm0));
        const ChP prm = (ChP)prm0;
        const int tid = wv * 64 + de_lane_here();
        const uint32_t list = prm->nstripes == 8u ? xcd : 0u;
        if (tid == 0) {
            uint32_t item = __hip_atomic_fetch_add((ChG)(prm->heads + list), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int p = 0, tx = 0, ty = 0;
            for (; p < 8; ++p) {
                const uint32_t x0 = prm->x0[p][list], w = prm->x0[p][list + 1] - x0, n = w * prm->tiles_y[p];
                if (item < n) { ty = (int)(item / w); tx = (int)(x0 + item % w); break; }
                item -= n;
            }
            bc[0] = p; bc[1] = tx; bc[2] = ty;
        }
        __syncthreads();
        const int p = __builtin_amdgcn_readfirstlane(bc[0]), tx = __builtin_amdgcn_readfirstlane(bc[1]), ty = __builtin_amdgcn_readfirstlane(bc[2]);
        if (p >= 8) break;

        // ---- wait for the previous direction's tiles within reach ------------------------------------------------
#ifndef CH_X_NOWAIT      /* timing experiment: no dependency waits (wrong results) */
        if (p > 0 && wv == 0) {
            const int q = p - 1;
            const ChDir me = CH_DIRS[p], pr = CH_DIRS[q];
            const int H = max(me.hu, pr.hu);
            const int by0 = ty * me.th;
            const int b0 = max(0, by0 - H) / pr.th, b1 = min((int)prm->d.ah - 1, by0 + me.th + H - 1) / pr.th;
            const int R = me.reach_x + pr.reach_x;
            const int xl = (tx * me.tw - R - pr.tw) / pr.tw, xh = (tx * me.tw + me.tw + R) / pr.tw;        // producer tiles' tx, before clipping
            const int txl = max(0, xl), txh = min((int)prm->tiles_x[q] - 1, xh);
            // lanes: (band, stripe) pairs.  nb bands x up to nstripes stripes, 64 per sweep
            const int nb = b1 - b0 + 1, ns = (int)prm->nstripes;
            bool ok = true;
            for (int base = 0; base < nb * ns; base += 64) {
                const int l = base + tid;
                const int b = b0 + l / ns, s = l % ns;
                const uint32_t x0 = prm->x0[q][s], x1 = prm->x0[q][s + 1];
                const bool need = l < nb * ns && x1 > x0 && (int)x0 <= txh && (int)x1 > txl && txl <= txh;
                const ChG cnt = (ChG)(prm->done + ((size_t)q * prm->maxbands + (uint32_t)b) * 8u + (uint32_t)s);
                uint32_t spins = 0;
                for (;;) {
                    const bool ready = !need || __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= x1 - x0;
                    if (__builtin_amdgcn_ballot_w64(!ready) == 0ull) break;
                    __builtin_amdgcn_s_sleep(8);
                    // ~a second: a broken dependency, not a slow neighbour.  Once one wait has given up every other one
                    // does at once (the launch then ends quickly, with a wrong image and the failure flag set)
                    if (++spins > (1u << 22) || ((spins & 1023u) == 0u && __hip_atomic_load((ChG)prm->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { ok = false; break; }
                }
                if (!ok) break;
            }
            if (!ok && tid == 0) __hip_atomic_store((ChG)prm->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
        __syncthreads();
        CH_T(1);

        {
            float4 *const ps = (p & 1) ? prm->img[1] : prm->img[0], *const pd = (p & 1) ? prm->img[0] : prm->img[1];
#ifdef CH_X_PLAIN        /* timing experiment: plain loads and stores (stale reads across XCDs) */
            typedef DeImgPlain CH_IMG;
            const CH_IMG src = {ps}, dst = {pd};
#else
            typedef DeImgSc1 CH_IMG;
            const CH_IMG src = {__builtin_amdgcn_make_buffer_rsrc((void *)ps, 0, 0x7fffffff, 0x00020000)};
            const CH_IMG dst = {__builtin_amdgcn_make_buffer_rsrc((void *)pd, 0, 0x7fffffff, 0x00020000)};
#endif
            switch (p) {
            case 0: ch_run<0, CH_IMG>(prm, src, dst, tx, ty, tid); break;
            case 1: ch_run<1, CH_IMG>(prm, src, dst, tx, ty, tid); break;
            case 2: ch_run<2, CH_IMG>(prm, src, dst, tx, ty, tid); break;
            case 3: ch_run<3, CH_IMG>(prm, src, dst, tx, ty, tid); break;
            case 4: ch_run<4, CH_IMG>(prm, src, dst, tx, ty, tid); break;
            case 5: ch_run<5, CH_IMG>(prm, src, dst, tx, ty, tid); break;
            case 6: ch_run<6, CH_IMG>(prm, src, dst, tx, ty, tid); break;
            default: ch_run<7, CH_IMG>(prm, src, dst, tx, ty, tid); break;
            }
        }
        CH_T(2);
        // ---- publish: every storing wave drains its write-through stores, then one lane counts the tile ------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (wv == 0 && de_lane_here() == 0) {
            const ChParams *pq0 = prm_;
            asm volatile("" : "+s"(pq0));
            const ChP pq = (ChP)pq0;
            const uint32_t s = pq->nstripes == 8u ? xcd : 0u;
            __hip_atomic_fetch_add((ChG)(pq->done + ((size_t)p * pq->maxbands + (uint32_t)ty) * 8u + s), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        CH_T(3);
#ifdef CH_X_TIMES
        if (threadIdx.x == 0) ch_times[blockIdx.x][4] += 1ull;
#endif
    }
}

// One direction per launch with the chain's tile shapes and arithmetic (plain loads and stores): what the persistent
// launch is compared with, bit for bit.
template <int P>
__global__ void __launch_bounds__(CH_NT, 8) k_de_one(const ChParams *__restrict__ q_, uint32_t ntiles)
{
    const ChP q = (ChP)q_;
    const uint32_t t = blockIdx.x;
    if (t >= ntiles) return;
    const DeImgPlain src = {q->img[P & 1]}, dst = {q->img[(P + 1) & 1]};
    ch_run<P, DeImgPlain>(q, src, dst, (int)(t % q->tiles_x[P]), (int)(t / q->tiles_x[P]), (int)threadIdx.x);
}

// ---- Overlapped launches (FLAME_DE_CHAIN=4): one launch per direction again, but direction p + 1 starts while the last
// round of direction p's workgroups is still running, and fills the compute units they leave.
//   * Directions alternate between two streams.  In front of direction p + 1 its stream runs k_de_gate, one wave that
//     waits until EVERY workgroup of direction p has started (hipExtAnyOrderLaunch is ignored on gfx950 and two kernels
//     simply co-dispatched could deadlock: waiting workgroups of p + 1 holding the slots p's remaining tiles need.
//     Once all of p's workgroups are resident, whatever p + 1 waits for is running, and by induction finishes).
//     tools/lap_probe.hip: the hand-over costs nothing measurable, hipStreamWaitValue64 does the same.
//   * Tiles in row-major order: p + 1 begins at the top of the image while p finishes the bottom, so the waits below
//     practically never spin.
//   * A tile of p + 1 waits for the flags of p's tiles within reach (same reach arithmetic as the persistent launch: read
//     after write for its staged region, write after read for its output, the two images ping-pong); hand-off as there:
//     write-through stores, L1-bypassing loads, drain + barrier, then the flag (cdna_hip_programming.md Guideline 16, R1).
// The started-workgroup count of a direction is spread over CH_NCNT words in separate 64-byte lines: agent-scope atomics
// on ONE address complete at ~18 M/s (8540 workgroups: 470 us — the first form of this kernel, and of the persistent launch's
// list heads, profiles/r04_de_lap.txt), and the workgroup's first wave waits for it with its loads (one in-order counter).
typedef __attribute__((address_space(1))) unsigned long long *ChG64;
constexpr int CH_NCNT = 256, CH_CNT_STRIDE = 8;
__global__ void k_de_gate(const ChParams *__restrict__ q_, int p, unsigned long long target)
{
    const ChP q = (ChP)q_;
    const ChG64 cnt = (ChG64)(q->started + (size_t)p * CH_NCNT * CH_CNT_STRIDE);
    uint32_t spins = 0;
    for (;;) {
        unsigned long long v = 0;
        for (int i = (int)threadIdx.x; i < CH_NCNT; i += 64) v += __hip_atomic_load(cnt + i * CH_CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, sh), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), sh);
            v += ((unsigned long long)hi << 32) | lo;
        }
        if (v >= target) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 23)) { if (threadIdx.x == 0) __hip_atomic_store((ChG)q->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
}

template <int P, class IMG>
__global__ void __launch_bounds__(CH_NT, 8) k_de_lap(const ChParams *__restrict__ q_, uint32_t ntiles)
{
    const ChP q = (ChP)q_;
    const uint32_t t = blockIdx.x;
    if (t >= ntiles) return;
    const uint32_t dbg = q->dbg;
    const int tiles_x = (int)q->tiles_x[P];
    const int tx = (int)(t % (uint32_t)tiles_x), ty = (int)(t / (uint32_t)tiles_x);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (P < 7 && threadIdx.x == 0 && !(dbg & 8u)) __hip_atomic_fetch_add((ChG64)(q->started + ((size_t)P * CH_NCNT + (t % (uint32_t)CH_NCNT)) * CH_CNT_STRIDE), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (P > 0 && !(dbg & 1u)) {
        if (wv == 0) {
            constexpr int Q = P > 0 ? P - 1 : 0;
            constexpr ChDir me = ch_dir<P>(), pr = ch_dir<Q>();
            constexpr int H = me.hu > pr.hu ? me.hu : pr.hu, R = me.reach_x + pr.reach_x;
            const int by0 = ty * me.th;
            const int b0 = max(0, by0 - H) / pr.th, b1 = min((int)q->d.ah - 1, by0 + me.th + H - 1) / pr.th;
            const int ptx = (int)q->tiles_x[Q];
            const int txl = max(0, (tx * me.tw - R - pr.tw) / pr.tw), txh = min(ptx - 1, (tx * me.tw + me.tw + R) / pr.tw);
            const int nx = txh - txl + 1, total = nx > 0 ? (b1 - b0 + 1) * nx : 0;
            const uint32_t epoch = q->epoch;
            const ChG flags = (ChG)(q->flags + (size_t)Q * q->maxtiles);
            const int lane = (int)threadIdx.x;
            bool ok = true;
            for (int base = 0; base < total && ok; base += 64) {
                const int l = base + lane;
                const bool need = l < total;
                const int b = b0 + (need ? l / nx : 0), x = txl + (need ? l % nx : 0);
                const ChG f = flags + (uint32_t)(b * ptx + x);
                uint32_t spins = 0;
                for (;;) {
                    const bool ready = !need || __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
                    if (__builtin_amdgcn_ballot_w64(!ready) == 0ull) break;
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1u << 22) || ((spins & 1023u) == 0u && __hip_atomic_load((ChG)q->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { ok = false; break; }
                }
            }
            if (!ok && lane == 0) __hip_atomic_store((ChG)q->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    {
        float4 *const ps = q->img[P & 1], *const pd = q->img[(P + 1) & 1];
        if constexpr (std::is_same<IMG, DeImgSc1>::value) {
            const DeImgSc1 src = {__builtin_amdgcn_make_buffer_rsrc((void *)ps, 0, 0x7fffffff, 0x00020000)};
            const DeImgSc1 dst = {__builtin_amdgcn_make_buffer_rsrc((void *)pd, 0, 0x7fffffff, 0x00020000)};
            ch_run<P, DeImgSc1>(q, src, dst, tx, ty, (int)threadIdx.x);
        } else {
            const DeImgPlain src = {ps}, dst = {pd};
            ch_run<P, DeImgPlain>(q, src, dst, tx, ty, (int)threadIdx.x);
        }
    }
    if (P < 7 && !(q->dbg & 2u)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (wv == 0 && de_lane_here() == 0) {
            const ChParams *pq0 = q_;
            asm volatile("" : "+s"(pq0));
            const ChP pq = (ChP)pq0;
            __hip_atomic_store((ChG)(pq->flags + (size_t)P * pq->maxtiles + t), pq->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct ChTiles { uint32_t x[8], y[8]; };
template <int P> void ch_tiles(const fl_dim &d, ChTiles &t)
{
    using G = DeGeo<P>;
    t.x[P] = (d.astride + G::SPAN + G::TW - 1) / G::TW;
    t.y[P] = (d.ah + G::TH - 1) / G::TH;
}

}      // namespace

static uint32_t ch_all_tiles(const fl_dim &d, ChTiles &t)
{
    ch_tiles<0>(d, t); ch_tiles<1>(d, t); ch_tiles<2>(d, t); ch_tiles<3>(d, t);
    ch_tiles<4>(d, t); ch_tiles<5>(d, t); ch_tiles<6>(d, t); ch_tiles<7>(d, t);
    uint32_t mb = 0;
    for (int p = 0; p < 8; ++p) mb = std::max(mb, t.y[p]);
    return mb;
}

size_t de_chain_scratch_bytes(fl_dim d)
{
    ChTiles t;
    return 1024 + (16 + (size_t)8 * ch_all_tiles(d, t) * 8) * sizeof(uint32_t);       // parameters, list heads, failure flag, counters
}

// The whole DE: img0 holds the input (in_mode as launch_de_dir's), the result lands in img0 again (eight passes).
// scratch: de_chain_scratch_bytes() of device memory.  one_by_one: eight launches of the same tiles.
static void ch_fill(ChParams &q, ChTiles &tl, fl_dim d, float4 *img0, float4 *img1, const float *coefs7, float sstd, float cstd, float dstd,
                    float dpow, float gspeed, int in_mode, const DeTail *tail)
{
    q.d = d; q.img[0] = img0; q.img[1] = img1;
    for (int i = 0; i < 7; ++i) q.kc.k[i] = coefs7[i];
    for (int m = -9; m <= 9; ++m) {
        float v = 0.0f;
        for (int i = -3; i <= 3; ++i) { const int j = m - 2 * i; if (j >= -3 && j <= 3) v += coefs7[i + 3] * coefs7[j + 3]; }
        q.kc.k2[m + 9] = v;
    }
    for (int r = 0; r < 16; ++r) q.spk.s[r] = expf((float)(r * r) / (-1.41421353816986f * sstd));
    q.cs2 = 1.0f / (-1.41421353816986f * 3.0f * cstd) * 1.44269502162933f;
    q.ads = fabsf(-0.5f / dstd);
    q.dpow = dpow; q.gspeed = gspeed;
    q.in_mode = in_mode; q.has_tail = tail ? 1 : 0;
    if (tail) q.tail = *tail;
    q.maxbands = ch_all_tiles(d, tl);
    for (int p = 0; p < 8; ++p) { q.tiles_x[p] = tl.x[p]; q.tiles_y[p] = tl.y[p]; }
}

void launch_de_chain(hipStream_t st, fl_dim d, float4 *img0, float4 *img1, const float *coefs7, float sstd, float cstd, float dstd,
                     float dpow, float gspeed, int in_mode, const DeTail *tail, void *scratch, bool one_by_one)
{
    ChParams q = {};
    ChTiles tl;
    ch_fill(q, tl, d, img0, img1, coefs7, sstd, cstd, dstd, dpow, gspeed, in_mode, tail);
    // one list per XCD when a stripe of the image is much wider than anything a tile reaches sideways
    uint32_t min_stripe_px = ~0u;
    int max_reach = 0;
    for (int p = 0; p < 8; ++p) {
        min_stripe_px = std::min(min_stripe_px, tl.x[p] / 8u * (uint32_t)CH_DIRS_H[p].tw);
        max_reach = std::max(max_reach, CH_DIRS_H[p].reach_x);
    }
    q.nstripes = min_stripe_px >= 3u * (uint32_t)max_reach ? 8u : 1u;
    for (int p = 0; p < 8; ++p) {
        for (uint32_t s_ = 0; s_ <= 8; ++s_) q.x0[p][s_] = s_ >= q.nstripes ? tl.x[p] : (uint32_t)((uint64_t)tl.x[p] * s_ / q.nstripes);
    }
    if (const char *e = getenv("FLAME_DE_CHAIN_ONLY")) {      // timing experiment: only this direction's tiles are in the lists (wrong results)
        const int only = atoi(e);
        for (int p = 0; p < 8; ++p) if (p != only) for (uint32_t s_ = 0; s_ <= 8; ++s_) q.x0[p][s_] = 0;
    }
    // scratch: the parameters (1 KB), the list heads, the failure flag, the counters
    static_assert(sizeof(ChParams) <= 1024, "parameter block");
    uint32_t *w = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(scratch) + 1024);
    q.heads = w; q.fail = w + 8; q.done = w + 16;
    ChParams *dq = static_cast<ChParams *>(scratch);
    hipLaunchKernelGGL(k_ch_params, dim3(1), dim3(64), 0, st, dq, q);
    if (one_by_one) {
        static unsigned long long attr[8] = {};
#define ONE(P) do { ensure_max_dynamic_lds((const void *)k_de_one<P>, attr[P]); const uint32_t n = tl.x[P] * tl.y[P]; \
                    hipLaunchKernelGGL(k_de_one<P>, dim3(n), dim3(CH_NT), DeGeo<P>::LDS, st, (const ChParams *)dq, n); } while (0)
        ONE(0); ONE(1); ONE(2); ONE(3); ONE(4); ONE(5); ONE(6); ONE(7);
#undef ONE
        return;
    }
    (void)hipMemsetAsync(w, 0, (16 + (size_t)8 * q.maxbands * 8) * sizeof(uint32_t), st);
    static unsigned long long attr = 0;
    ensure_max_dynamic_lds((const void *)k_de_chain, attr);
    hipLaunchKernelGGL(k_de_chain, dim3(256 * 8), dim3(CH_NT), CH_LDS + 64, st, (const ChParams *)dq);
}

// ---- overlapped launches -------------------------------------------------------------------------------------------
static uint32_t lap_maxtiles(fl_dim d)
{
    ChTiles t; ch_all_tiles(d, t);
    uint32_t m = 0;
    for (int p = 0; p < 8; ++p) m = std::max(m, t.x[p] * t.y[p]);
    return m;
}
// scratch: [0, 1024) parameters | [1056] failure flag (where the persistent launch has it) | [2048, +128 KB) started counts | flags
static const size_t LAP_STARTED = 2048, LAP_FLAGS = LAP_STARTED + (size_t)8 * CH_NCNT * CH_CNT_STRIDE * sizeof(unsigned long long);
size_t de_lap_scratch_bytes(fl_dim d) { return LAP_FLAGS + (size_t)8 * lap_maxtiles(d) * sizeof(uint32_t); }

// sa: the lane's stream (directions 0, 2, 4, 6), sb: a second one (1, 3, 5, 7); sb is ordered behind sa's earlier work through
// `fork`, sa waits for the last direction through `join`.  state: epoch and started-workgroup totals of THIS scratch (all zero
// after the scratch was allocated and cleared).
void launch_de_lap(hipStream_t sa, hipStream_t sb, hipEvent_t fork, hipEvent_t join, fl_dim d, float4 *img0, float4 *img1, const float *coefs7,
                   float sstd, float cstd, float dstd, float dpow, float gspeed, int in_mode, const DeTail *tail, void *scratch, DeLapState *state)
{
    ChParams q = {};
    ChTiles tl;
    ch_fill(q, tl, d, img0, img1, coefs7, sstd, cstd, dstd, dpow, gspeed, in_mode, tail);
    q.nstripes = 1;
    unsigned char *base = static_cast<unsigned char *>(scratch);
    q.fail = reinterpret_cast<uint32_t *>(base + 1056);
    q.started = reinterpret_cast<unsigned long long *>(base + LAP_STARTED);
    q.flags = reinterpret_cast<uint32_t *>(base + LAP_FLAGS);
    q.maxtiles = lap_maxtiles(d);
    q.epoch = ++state->epoch;
    static const uint32_t dbg = getenv("FLAME_DE_LAP_DBG") ? (uint32_t)atoi(getenv("FLAME_DE_LAP_DBG")) : 0u;
    q.dbg = dbg;
    if (dbg & 4u) sb = sa;
    ChParams *dq = static_cast<ChParams *>(scratch);
    hipLaunchKernelGGL(k_ch_params, dim3(1), dim3(64), 0, sa, dq, q);
    if (sb != sa) { (void)hipEventRecord(fork, sa); (void)hipStreamWaitEvent(sb, fork, 0); }
    static unsigned long long attr[8] = {}, attr_pl[8] = {};
#define LAP(P) do { const uint32_t n = tl.x[P] * tl.y[P]; hipStream_t st_ = (P & 1) ? sb : sa; \
                    if (P > 0 && sb != sa) hipLaunchKernelGGL(k_de_gate, dim3(1), dim3(64), 0, st_, (const ChParams *)dq, P - 1, state->started[P > 0 ? P - 1 : 0]); \
                    if (dbg & 16u) { ensure_max_dynamic_lds((const void *)k_de_lap<P, DeImgPlain>, attr_pl[P]); \
                                     hipLaunchKernelGGL((k_de_lap<P, DeImgPlain>), dim3(n), dim3(CH_NT), DeGeo<P>::LDS, st_, (const ChParams *)dq, n); } \
                    else { ensure_max_dynamic_lds((const void *)k_de_lap<P, DeImgSc1>, attr[P]); \
                           hipLaunchKernelGGL((k_de_lap<P, DeImgSc1>), dim3(n), dim3(CH_NT), DeGeo<P>::LDS, st_, (const ChParams *)dq, n); } \
                    if (P < 7 && !(dbg & 8u)) state->started[P] += n; } while (0)
    LAP(0); LAP(1); LAP(2); LAP(3); LAP(4); LAP(5); LAP(6); LAP(7);
#undef LAP
    if (sb != sa) { (void)hipEventRecord(join, sb); (void)hipStreamWaitEvent(sa, join, 0); }
}

// 1 when a workgroup of the last persistent launch gave up waiting for a neighbour (a bug; tests look at it)
int de_chain_failed(const void *scratch)
{
    uint32_t f = 0;
    if (!scratch || hipMemcpy(&f, static_cast<const unsigned char *>(scratch) + 1024 + 8 * sizeof(uint32_t), sizeof f, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return f != 0;
}

#ifdef CH_X_TIMES
extern "C" __attribute__((visibility("default"))) int fl_debug_chain_times(unsigned long long *out, int clear)
{
    static unsigned long long host[2048][5];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ch_times), sizeof host) != hipSuccess) return -1;
    for (int n = 0; n < 5; ++n) { unsigned long long a = 0; for (int w = 0; w < 2048; ++w) a += host[w][n]; out[n] = a; }
    if (clear) { void *sym = nullptr; if (hipGetSymbolAddress(&sym, HIP_SYMBOL(ch_times)) != hipSuccess || hipMemset(sym, 0, sizeof host) != hipSuccess) return -1; }
    return 0;
}
#endif
