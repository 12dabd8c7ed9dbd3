// sort.hip — stable LSD radix sort of 32-bit keys, one pass per call (1..10 bits at any position).
//
// Role of cuburn/code/sort.py (Sorter.sort, :443-504; kernels :33-382): the reference sorts by one
// radix digit per pass in groups of 8192 keys with a prefix-scan / condense / distribute kernel
// chain; its single pass "is not quite stable", which breaks its multi-pass sort (:437-441, :455-458).
// This one is stable, so passes compose into a full sort.  Nothing on the render path calls it (the
// render path's sort is the in-LDS tile sort of iter.hip) — same status as in the reference
// (imported by render.py:19, never called).
//
// MI355X mapping: HBM-bound integer work, three kernels per pass.
//   k_sort_hist     a workgroup (256 threads) counts the digits of its tile of 4096 keys with LDS
//                   atomics and stores them digit-major: hist[digit][tile]
//   k_sort_scan*    exclusive scan of hist in place (chunks of 4096 entries + one workgroup over the
//                   chunk totals): hist[digit][tile] becomes the global position of that tile's first
//                   key with that digit
//   k_sort_scatter  the tile's keys are ranked (wave-level match by ballots, waves and rounds chained
//                   through LDS, which keeps the order of equal digits), reordered in LDS, and written
//                   so that consecutive lanes write consecutive addresses of a digit's segment
// Algorithmic traffic per pass: 4 B read (hist) + 4 B read + 4 B written (scatter) = 12 B per key.
#include "flame_device.h"
#include "kernels.h"

#define SORT_TILE 4096u
#define SORT_THREADS 256u
#define SORT_KPT (SORT_TILE / SORT_THREADS)     /* keys per thread */
#define SORT_MAX_BITS 10u

__device__ __forceinline__ uint32_t sort_digit(uint32_t key, uint32_t lo_bit, uint32_t mask) { return (key >> lo_bit) & mask; }

__global__ void __launch_bounds__(SORT_THREADS)
k_sort_hist(const uint32_t *__restrict__ src, uint32_t n, uint32_t lo_bit, uint32_t nbits, int ignore_max,
            uint32_t *__restrict__ hist, uint32_t ntiles)
{
    __shared__ uint32_t cnt[1u << SORT_MAX_BITS];
    const uint32_t radix = 1u << nbits, mask = radix - 1u, tile = blockIdx.x, tid = threadIdx.x;
    for (uint32_t i = tid; i < radix; i += SORT_THREADS) cnt[i] = 0;
    __syncthreads();
    const uint32_t base = tile * SORT_TILE;
#pragma unroll
    for (uint32_t k = 0; k < SORT_KPT; ++k) {
        const uint32_t i = base + k * SORT_THREADS + tid;
        if (i < n) {
            const uint32_t key = src[i];
            if (!(ignore_max && key == 0xffffffffu))
                __hip_atomic_fetch_add(cnt + sort_digit(key, lo_bit, mask), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    for (uint32_t d = tid; d < radix; d += SORT_THREADS) hist[(size_t)d * ntiles + tile] = cnt[d];
}

__device__ __forceinline__ uint32_t sort_wave_incl_scan(uint32_t v)      // DPP scan (see iter.hip)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// exclusive scan of a workgroup's 256 * PER values (thread t holds values t*PER .. t*PER+PER-1 of the
// chunk); returns the chunk total
template <uint32_t PER>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t (&v)[PER], uint32_t *wave_tot /* [4] LDS */)
{
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) { const uint32_t t = v[k]; v[k] = sum; sum += t; }
    const uint32_t incl = sort_wave_incl_scan(sum);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (lane == 63u) wave_tot[w] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (uint32_t i = 0; i < SORT_THREADS / 64u; ++i) { const uint32_t t = wave_tot[i]; if (i < w) base += t; total += t; }
    base += incl - sum;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) v[k] += base;
    __syncthreads();
    return total;
}

__global__ void __launch_bounds__(SORT_THREADS)
k_sort_scan_chunks(uint32_t *__restrict__ a, uint32_t m, uint32_t *__restrict__ chunk_tot)
{
    __shared__ uint32_t wt[SORT_THREADS / 64u];
    const uint32_t base = blockIdx.x * SORT_TILE + threadIdx.x * SORT_KPT;
    uint32_t v[SORT_KPT];
#pragma unroll
    for (uint32_t k = 0; k < SORT_KPT; ++k) v[k] = base + k < m ? a[base + k] : 0u;
    const uint32_t total = block_excl_scan<SORT_KPT>(v, wt);
#pragma unroll
    for (uint32_t k = 0; k < SORT_KPT; ++k) if (base + k < m) a[base + k] = v[k];
    if (threadIdx.x == 0) chunk_tot[blockIdx.x] = total;
}

// one workgroup: exclusive scan of the chunk totals in place; the grand total goes to *total_out
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_scan_totals(uint32_t *__restrict__ t, uint32_t nchunks, uint32_t *__restrict__ total_out)
{
    __shared__ uint32_t wt[SORT_THREADS / 64u];
    uint32_t running = 0;
    for (uint32_t c0 = 0; c0 < nchunks; c0 += SORT_TILE) {
        const uint32_t base = c0 + threadIdx.x * SORT_KPT;
        uint32_t v[SORT_KPT];
#pragma unroll
        for (uint32_t k = 0; k < SORT_KPT; ++k) v[k] = base + k < nchunks ? t[base + k] : 0u;
        const uint32_t total = block_excl_scan<SORT_KPT>(v, wt);
#pragma unroll
        for (uint32_t k = 0; k < SORT_KPT; ++k) if (base + k < nchunks) t[base + k] = v[k] + running;
        running += total;
    }
    if (threadIdx.x == 0) *total_out = running;
}

__global__ void __launch_bounds__(SORT_THREADS)
k_sort_scatter(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint32_t n, uint32_t lo_bit, uint32_t nbits,
               int ignore_max, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ chunk_pfx, uint32_t ntiles)
{
    constexpr uint32_t NW = SORT_THREADS / 64u, RMAX = 1u << SORT_MAX_BITS;
    __shared__ uint32_t keys[SORT_TILE];            // the tile in sorted order
    __shared__ uint32_t start[RMAX];                // first local position of a digit; later: global - local
    __shared__ uint32_t run[RMAX];                  // next free local position of a digit (rounds so far)
    __shared__ uint32_t wcnt[NW][RMAX];             // keys of a digit per wave in the current round
    __shared__ uint32_t wt[NW];
    const uint32_t radix = 1u << nbits, mask = radix - 1u, tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t lane = tid & 63u, w = tid >> 6;
    const uint32_t base = tile * SORT_TILE;
    const unsigned long long lt = (1ull << lane) - 1ull;

    uint32_t key[SORT_KPT];
    bool ok[SORT_KPT];
    for (uint32_t i = tid; i < radix; i += SORT_THREADS) run[i] = 0;
    for (uint32_t i = tid; i < NW * RMAX; i += SORT_THREADS) (&wcnt[0][0])[i] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < SORT_KPT; ++k) {
        const uint32_t i = base + k * SORT_THREADS + tid;
        key[k] = i < n ? src[i] : 0xffffffffu;
        ok[k] = i < n && !(ignore_max && key[k] == 0xffffffffu);
        if (ok[k]) __hip_atomic_fetch_add(run + sort_digit(key[k], lo_bit, mask), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    // local exclusive scan of the tile's digit counts (radix <= 1024 = 256 threads x 4)
    {
        uint32_t v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) v[k] = tid * 4 + k < radix ? run[tid * 4 + k] : 0u;
        (void)block_excl_scan<4>(v, wt);
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) if (tid * 4 + k < radix) { start[tid * 4 + k] = v[k]; run[tid * 4 + k] = v[k]; }
    }
    __syncthreads();
    // stable ranks, one round of 256 keys at a time: order inside the tile = (round, wave, lane)
#pragma unroll
    for (uint32_t k = 0; k < SORT_KPT; ++k) {
        const uint32_t d = sort_digit(key[k], lo_bit, mask);
        // lanes of this wave holding the same digit (one ballot per digit bit)
        unsigned long long peers = __ballot(ok[k]);
        for (uint32_t b = 0; b < nbits; ++b) {
            const unsigned long long has = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? has : ~has;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lt), cntw = (uint32_t)__popcll(peers);
        const bool leader = ok[k] && rank == 0u;
        if (leader) wcnt[w][d] = cntw;
        __syncthreads();
        uint32_t pos = 0;
        if (ok[k]) {
            pos = run[d] + rank;
            for (uint32_t i = 0; i < w; ++i) pos += wcnt[i][d];
        }
        __syncthreads();
        if (leader) {                               // one leader per (wave, digit): chain the waves' counts into run[]
            __hip_atomic_fetch_add(run + d, cntw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            wcnt[w][d] = 0;
        }
        if (ok[k]) keys[pos] = key[k];
        __syncthreads();
    }
    // global position of a digit's first key of this tile, minus its local position
    for (uint32_t d = tid; d < radix; d += SORT_THREADS) {
        const size_t e = (size_t)d * ntiles + tile;
        start[d] = hist[e] + chunk_pfx[e / SORT_TILE] - start[d];
    }
    const uint32_t nvalid = run[radix - 1u];        // after the rounds: end of the last digit = number of keys kept
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < SORT_KPT; ++k) {
        const uint32_t i = k * SORT_THREADS + tid;
        if (i < nvalid) {
            const uint32_t kv = keys[i];
            dst[start[sort_digit(kv, lo_bit, mask)] + i] = kv;
        }
    }
}

int launch_sort_pass(hipStream_t st, uint32_t *dst, const uint32_t *src, uint32_t n, uint32_t lo_bit, uint32_t nbits,
                     int ignore_max, uint32_t *hist, uint32_t *chunk_tot, uint32_t *total_dev)
{
    const uint32_t ntiles = (n + SORT_TILE - 1) / SORT_TILE, radix = 1u << nbits;
    const uint32_t m = radix * ntiles, nchunks = (m + SORT_TILE - 1) / SORT_TILE;
    hipLaunchKernelGGL(k_sort_hist, dim3(ntiles), dim3(SORT_THREADS), 0, st, src, n, lo_bit, nbits, ignore_max, hist, ntiles);
    hipLaunchKernelGGL(k_sort_scan_chunks, dim3(nchunks), dim3(SORT_THREADS), 0, st, hist, m, chunk_tot);
    hipLaunchKernelGGL(k_sort_scan_totals, dim3(1), dim3(SORT_THREADS), 0, st, chunk_tot, nchunks, total_dev);
    hipLaunchKernelGGL(k_sort_scatter, dim3(ntiles), dim3(SORT_THREADS), 0, st, dst, src, n, lo_bit, nbits, ignore_max, hist, chunk_tot, ntiles);
    return 0;
}

size_t sort_scratch_words(uint32_t n, uint32_t nbits, size_t *chunk_words)
{
    const size_t ntiles = ((size_t)n + SORT_TILE - 1) / SORT_TILE, m = ((size_t)1 << nbits) * ntiles;
    *chunk_words = (m + SORT_TILE - 1) / SORT_TILE + 1;      // + the grand total
    return m;
}
