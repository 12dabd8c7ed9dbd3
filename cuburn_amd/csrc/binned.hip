// binned.hip — second half of the binned accumulate (FL_ACCUM_BINNED).
//
// k_iter (ACC == 1) leaves, per launch, a sample log: for every (slot, batch) a block of
// 21-bit records sorted by image tile — three to an aligned 64-bit word (flame_device.h, FL_LOG_PACK3;
// the 22-bit records of 256x64 tiles: one per 32-bit word) — plus a directory
// dir[tile][batch] = (first record << 16) | count.
// k_accum_tiles gives each 128x64-pixel tile to P workgroups (contiguous ranges of batches);
// a workgroup accumulates its share into an LDS tile of packed 64-bit cells with LDS atomics,
// then adds the tile to the global packed accumulator with COALESCED 64-bit atomics (64
// consecutive cells per wave instruction: 7x the throughput of scattered atomics on MI355X,
// profiles/r01_atomic_microbench*.txt).  Cells that fill up (>= 512 hits) are drained into the
// float accumulator exactly as in the direct path (cuburn/code/iter.py:366-406).
//
// The kernel is latency bound (directory -> run lookup -> record -> palette -> LDS atomic is a
// dependent chain per wave, the records come from HBM ~2400 clocks after they are asked for), so:
// 64 KB tiles (two 1024-thread workgroups = 32 waves per CU: 64 VGPRs and — measured — 80 SGPRs per
// wave at most), and a wave walks its runs in steps of ILP x 64 log words (a word = three records, or one), two steps
// in flight: the loads are issued from inline asm and waited for with partial s_waitcnt, so that step n+1
// is on its way while step n goes through the palette and the tile (tools/check_asm_atomics.py
// checks the assembly for what the compiler cannot know about those registers; DESIGN.md 4.1
// "Round 3 (second half)" has the measurements behind every piece of this).
#include "flame_device.h"
#include "kernels.h"
#include <cstdlib>

__device__ __forceinline__ void spill_cell(u64 cur, uint32_t gi, float *__restrict__ out4)
{
    float yf, uf, vf, df;
    unpack_cell(cur, yf, uf, vf, df);
    float *o = out4 + 4 * (size_t)gi;
    __hip_atomic_fetch_add(o + 0, yf * FL_INV255, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(o + 1, uf * FL_INV255, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(o + 2, vf * FL_INV255, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(o + 3, df, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum of the packed values of the lanes in `grp`, spilled to the float accumulator by the group's first lane
#ifndef HOT_GROUP_INLINE
#define HOT_GROUP_INLINE __forceinline__      /* a real call in the record loop keeps everything that lives across it in the few callee-saved VGPRs: spills */
#endif
__device__ HOT_GROUP_INLINE void hot_group(u64 v, unsigned long long grp, uint32_t gi, float *__restrict__ out4)
{
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, sh), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), sh);
        v += ((u64)hi << 32) | lo;
    }
    if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(grp)) spill_cell(v, gi, out4);
}

// the same for k_accum_tiles_p3
__device__ HOT_GROUP_INLINE void hot_group_sw(u64 v, unsigned long long grp, uint32_t gi, float *__restrict__ out4)
{
    // (xor 1 .. 16 as ds_swizzle with the pattern in the instruction: the lane addresses of six __shfl_xor are loop-invariant, the
    // compiler computes them in front of the record loop and keeps — or spills — five registers for a path that is almost never taken)
#define HOT_XOR(sh) do { const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(uint32_t)v, 0x1f | ((sh) << 10)), \
                                         hi = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(uint32_t)(v >> 32), 0x1f | ((sh) << 10)); \
                         v += ((u64)hi << 32) | lo; } while (0)
    HOT_XOR(1); HOT_XOR(2); HOT_XOR(4); HOT_XOR(8); HOT_XOR(16);
#undef HOT_XOR
    {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, 32), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), 32);
        v += ((u64)hi << 32) | lo;
    }
    if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(grp)) spill_cell(v, gi, out4);
}

__device__ __forceinline__ uint32_t wave_incl_scan_b(uint32_t v, uint32_t) {      // DPP scan, see iter.hip
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// Wave-level synchronisation point for the compiler (no instruction: a wave's LDS operations
// execute in order).  Lanes of one wave communicating through LDS are, to the compiler, distinct
// threads racing on plain memory; the fences make the exchange well-defined.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_incl_maxscan(uint32_t v) {                // same DPP steps, max instead of add
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}

#ifndef ACC_ILP_WIDE
#define ACC_ILP_WIDE 4     /* records per lane in flight, 256x64 tiles (one 16-wave workgroup per CU) */
#endif
#ifndef ACC_ILP
#define ACC_ILP 3          /* records per lane and step, 128x64 tiles: two steps are in flight (ACC_PIPE), and four per step do not fit 64 VGPRs without spills */
#endif
#ifndef ACC_GATHER_AHEAD
#define ACC_GATHER_AHEAD 4     /* palette entries requested this many records ahead of their add (4: all of a step's at once; fewer: fewer VGPRs) */
#endif
#ifndef ACC_DIR_AHEAD
#define ACC_DIR_AHEAD 1        /* a group's directory words are requested one group ahead (the first group's before the tile is zeroed) */
#endif
#ifndef ACC_BYTE_MARKS
#define ACC_BYTE_MARKS 1       /* run lookup: one LDS exchange of byte marks per step + ds_bpermute, instead of one exchange of word marks per 64 records */
#endif
#ifndef ACC_LOAD_MOD
#define ACC_LOAD_MOD ""        /* cache policy of the record loads (" nt", " sc1", ...): experiment, see profiles/r03_accum_cache_policy.txt */
#endif
#ifndef ACC_PIPE
#define ACC_PIPE 1             /* the next step's record loads are in flight while the current step's records are added */
#endif
#ifndef ACC_ADD_ILP
#define ACC_ADD_ILP 8          /* returning global atomics in flight per thread when the tile is added to the accumulator */
#endif
#ifndef ACC_ROWS_MAX
#define ACC_ROWS_MAX 5         /* palette rows staged per narrow workgroup (2 KB each) */
#endif
#ifndef ACC_THREADS
#define ACC_THREADS 1024        /* threads per accumulate workgroup (narrow tiles) */
#endif

// -DACC_X_TIMES: every workgroup stores its start and end (100 MHz ticks) in acc_wg_times[blockIdx.x]; fl_debug_acc_times
// copies them out (tools/acc_times.py: which tiles' workgroups run longest, and when)
#ifdef ACC_X_TIMES
#define ACC_X_MAXWG 65536
__device__ unsigned long long acc_wg_times[ACC_X_MAXWG][4];      // start, tile zeroed + first palette rows staged, records done, tile added
// wave 0 of every workgroup also sums the shader clocks (s_memtime) it spends requesting a step's records (run lookup:
// the mark exchange), waiting for a step's records, and adding them; [3] = the whole record phase, [4] = steps
__device__ unsigned long long acc_wg_steps[ACC_X_MAXWG][5];
extern "C" __attribute__((visibility("default"))) int fl_debug_acc_steps(unsigned long long *out, unsigned n)
{
    if (n > ACC_X_MAXWG) n = ACC_X_MAXWG;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(acc_wg_steps), sizeof(unsigned long long) * 5 * n) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int fl_debug_acc_times(unsigned long long *out, unsigned n)
{
    if (n > ACC_X_MAXWG) n = ACC_X_MAXWG;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(acc_wg_times), sizeof(unsigned long long) * 4 * n) == hipSuccess ? 0 : -1;
}
#endif

// The end of every accumulate workgroup: its LDS tile is added to the global packed accumulator, one row segment of 64 cells
// per wave instruction (coalesced atomics), draining cells that reach 512 hits.
template <uint32_t TWL>
__device__ __forceinline__ void add_tile_to_cells(const u64 *tile, u64 *__restrict__ atom, float *__restrict__ out4,
                                                  uint32_t tx, uint32_t ty, uint32_t astride, uint32_t aheight, uint32_t big_thr)
{
    constexpr uint32_t TW = 1u << TWL, CELLS = TW * FL_TILE_H;
    const uint32_t tid = threadIdx.x;
    // ACC_ADD_ILP returning atomics per thread are in flight before the first result is looked at (one at a time,
    // the loop was eight serial round trips to L2: 6.5 us of a 43 us workgroup)
    for (uint32_t i0 = tid; i0 < CELLS; i0 += blockDim.x * ACC_ADD_ILP) {
        u64 old[ACC_ADD_ILP];
        uint32_t big = 0;
#pragma unroll
        for (uint32_t k = 0; k < ACC_ADD_ILP; ++k) {
            const uint32_t i = i0 + k * blockDim.x;
            old[k] = 0ull;
            if (i < CELLS) {
                const u64 v = tile[i];
                const uint32_t px = tx * TW + (i & (TW - 1u)), py = ty * FL_TILE_H + (i >> TWL);
                if (v != 0ull && px < astride && py < aheight) {
                    // A packed add must never carry the 10-bit count past 1023.  Each cell receives at
                    // most `nparts` adds per launch (one per workgroup of its tile) onto a flushed cell,
                    // so chunks below big_thr = 1024/nparts hits are safe; larger ones go straight to the floats.
                    if ((uint32_t)(v >> 54) >= big_thr) big |= 1u << k;
                    else                                         // (as asm: the compiler waits for every returning atomic at the end of its branch)
                        asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0" : "+v"(old[k]) : "v"(atom + py * astride + px), "v"(v) : "memory");
                }
            }
        }
        static_assert(ACC_ADD_ILP == 8 || ACC_ADD_ILP == 4, "the wait below names every result");
        if constexpr (ACC_ADD_ILP == 8)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(old[4]), "+v"(old[5]), "+v"(old[6]), "+v"(old[7]) :: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]) :: "memory");
#pragma unroll
        for (uint32_t k = 0; k < ACC_ADD_ILP; ++k) {
            const bool full = (uint32_t)(old[k] >> 32) >= (256u << 23);
            if (full || ((big >> k) & 1u)) {                      // rare: the cell is looked up again
                const uint32_t i = i0 + k * blockDim.x;
                const uint32_t gi = (ty * FL_TILE_H + (i >> TWL)) * astride + tx * TW + (i & (TW - 1u));
                if (full) {
                    const u64 cur = __hip_atomic_exchange(atom + gi, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(cur >> 32) != 0u) spill_cell(cur, gi, out4);
                } else
                    spill_cell(tile[i], gi, out4);
            }
        }
    }
}

// One record per 32-bit log word: 256x64 tiles, and 128x64 tiles of a build without FL_LOG_PACK3 (the packed log has its own kernel below).
// This is round 5's kernel, unchanged (a round-6 refactoring of it for a templated packed form measured +4.8 % at 8K and was taken back).
// TWL: log2 of the tile width (7: 128x64 tiles = 64 KB of LDS cells, two workgroups per CU;
// 8: 256x64 tiles = 128 KB, for images with more than 2047 narrow tiles)
// 80 SGPRs including VCC etc.: two 16-wave workgroups per CU need 8 waves per SIMD, and a CU of this GPU holds
// 8 waves per SIMD only up to 80 SGPRs per wave (measured: tools/occupancy_probe.hip,
// profiles/r03_occupancy_probe.txt; the compiler's table and the occupancy API say 96)
template <uint32_t TWL>
__global__ void __launch_bounds__(TWL == 7u ? ACC_THREADS : 1024, TWL == 7u ? 8 : 4) __attribute__((amdgpu_num_sgpr(80)))
k_accum_tiles(const uint32_t *__restrict__ log, const uint32_t *__restrict__ dir,
              const u64 *__restrict__ palette, u64 *__restrict__ atom, float *__restrict__ out4,
              uint32_t tiles_x, uint32_t nparts, uint32_t nbatch_total, uint32_t batch_records,
              uint32_t nslots, uint32_t astride, uint32_t aheight, uint32_t rows_cap, uint32_t big_thr, uint32_t gang, uint32_t nbins)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t TW = 1u << TWL, CELLS = TW * FL_TILE_H;
    constexpr int ILP = TWL == 7u ? ACC_ILP : ACC_ILP_WIDE;
    static_assert(FL_PAL_W == 256, "the palette column is the record's low byte, the row the mark's");
    // LDS: palette rows first (their gather then needs no base added), the waves' marks, the tile
    u64 *pal = reinterpret_cast<u64 *>(smem);                                                   // [rows_cap][256] palette rows in use
    uint32_t *mk = reinterpret_cast<uint32_t *>(smem + rows_cap * FL_PAL_W * 8) + (threadIdx.x >> 6) * 64;   // [64] marks of this wave
    u64 *tile = reinterpret_cast<u64 *>(smem + rows_cap * FL_PAL_W * 8 + blockDim.x * 4);       // [CELLS]
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwaves = blockDim.x >> 6;
    // Workgroup -> (tile, part).  gang = 0: parts of a tile are consecutive workgroups.  gang = G (round 5 experiment, FLAME_BIN_GANG):
    // G adjacent tiles with the SAME part are G consecutive workgroups of ONE XCD (workgroup b runs on XCD b % 8) — they start
    // together on one L2 and walk the same batches, whose sorted records put adjacent tiles' runs into the same cache lines.
    uint32_t bin, part;
    if (gang == 0u) { bin = blockIdx.x / nparts; part = blockIdx.x % nparts; }
    else {
        const uint32_t loc = blockIdx.x >> 3, gq = (loc / gang) * 8u + (blockIdx.x & 7u);
        bin = (gq / nparts) * gang + loc % gang; part = gq % nparts;
        if (bin >= nbins) return;
    }
    const uint32_t tx = bin % tiles_x, ty = bin / tiles_x;
#ifdef ACC_X_TIMES
    if (tid == 0 && blockIdx.x < ACC_X_MAXWG) acc_wg_times[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
#endif


    // This workgroup's contiguous range of batches.  Batch id = slot * per_slot + batch_in_slot
    // (iter.hip), so the range covers a narrow range of SLOTS, and with them of palette rows (row of
    // a batch = its slot * 64 / nslots): the rows in use are staged in LDS — the palette fetch is a
    // dependent, fully scattered 8-byte gather, a quarter of this kernel's time from L2, a few
    // cycles from LDS.  Ranges that need more than rows_cap rows are walked in chunks of
    // (rows_cap - 1) * slots_per_row slots, re-staging in between.
    const uint32_t b_lo = (uint32_t)((u64)nbatch_total * part / nparts);
    const uint32_t b_hi = (uint32_t)((u64)nbatch_total * (part + 1) / nparts);
    const uint32_t *drow = dir + (size_t)bin * nbatch_total;
    const uint32_t per_slot = nbatch_total / nslots;
    const uint32_t spr = nslots / FL_PAL_H;                       // slots per palette row
    // (wave-uniform, but computed by the vector ALU: moved to scalar registers so that they do not hold two VGPRs through the loops)
    const float inv_spr = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (float)spr)));
    const float inv_ps = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (float)per_slot)));
    const uint32_t chunk_slots = (rows_cap - 1u) * spr;

    if (b_lo >= b_hi) return;                                      // no batches for this part (tiny launches): nothing to add
#ifdef ACC_X_TIMES
    uint32_t x_fetch = 0, x_wait = 0, x_proc = 0, x_other = 0, x_steps = 0;
    const uint32_t x_t0 = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    for (uint32_t cb = b_lo; cb < b_hi;) {
    const uint32_t cs_lo = cb / per_slot;
    const uint32_t ce = min(b_hi, (cs_lo + chunk_slots) * per_slot);
    const uint32_t row_lo = cs_lo / spr, nrows = min(rows_cap, (uint32_t)FL_PAL_H - row_lo);
    // (the first chunk's rows are requested before the tile is zeroed: the workgroup's first global
    // latency then runs under the zeroing instead of after it)
    u64 stagev[3];                                                 // rows_cap * 256 <= 3 * blockDim.x
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const uint32_t i = tid + q * blockDim.x;
        stagev[q] = i < nrows * FL_PAL_W ? palette[row_lo * FL_PAL_W + i] : 0ull;
    }
    // (so are the directory words of the wave's first group; every group then requests the next group's words
    // before it walks its own records)
#if ACC_DIR_AHEAD
    uint32_t e_next = cb + wv * 64 + lane < ce ? drow[cb + wv * 64 + lane] : 0u;
#endif
    if (cb == b_lo) {
        for (uint32_t i = tid; i < CELLS; i += blockDim.x) tile[i] = 0ull;
        mk[lane] = 0u;
    }
    __syncthreads();                                               // readers of the previous chunk's rows are done
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const uint32_t i = tid + q * blockDim.x;
        if (i < nrows * FL_PAL_W) pal[i] = stagev[q];
    }
    __syncthreads();
#ifdef ACC_X_TIMES
    if (tid == 0 && blockIdx.x < ACC_X_MAXWG && cb == b_lo) acc_wg_times[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
#endif
    for (uint32_t g0v = cb + wv * 64; g0v < ce; g0v += nwaves * 64) {
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g0v);      // wave-uniform
        // slot of the group's first batch and the remainder, once per group (scalar); run r < 64 of
        // the group then belongs to slot s0 + (rem0 + r) / per_slot
        const uint32_t s0 = g0 / per_slot, rem0 = g0 - s0 * per_slot;
        // 64 directory entries per wave; their runs form one virtual array of `total` records
        const uint32_t batch = g0 + lane;
#if ACC_DIR_AHEAD
        const uint32_t e = e_next;
        e_next = batch + nwaves * 64 < ce ? drow[batch + nwaves * 64] : 0u;
#else
        const uint32_t e = batch < ce ? drow[batch] : 0u;
#endif
        const uint32_t c = e & 0xffffu, first = e >> 16;
        const uint32_t incl = wave_incl_scan_b(c, lane);
        const uint32_t excl = incl - c;
        const uint32_t total = __shfl(incl, 63);
        // Which run does record v of the virtual array belong to?  Every non-empty run drops a mark at its first
        // position; a max-scan over the positions (DPP, pure VALU) then carries the latest mark to every position.
        // This replaces a 6-step shuffle binary search + two more shuffles per record (8 trips through the LDS pipe)
        // by one predicated LDS write, one read and one clear per 64 records.  The mark holds everything a record
        // needs from its run, computed ONCE per run by the run's directory lane:
        //   bits 8..: 1 + (lane * batch_records + first - excl) = 1 + (the record's index in the log, counted from
        //             the group's first batch) - (its position v in the virtual array)
        //   bits 0-7: the run's palette row among the staged rows (a whole byte: v_perm joins it with the record's colour byte)
        // The upper field never decreases from one run to the next (first' + batch_records >= first + c: a run ends
        // inside its batch) and is the same only where the record index is the same anyway; the row never decreases
        // either, so the LARGEST mark at or before a position is the mark of the run the position belongs to.
        const uint32_t rslot_l = s0 + (uint32_t)(((float)(rem0 + lane) + 0.5f) * inv_ps);          // exact: small integers
        const uint32_t row_l = (uint32_t)(((float)rslot_l + 0.5f) * inv_spr) - row_lo;              // row within the staged rows
        const uint32_t mark_l = ((lane * batch_records + first - excl + 1u) << 8) | (row_l & 255u);
        const unsigned char *gbase = reinterpret_cast<const unsigned char *>(log + (size_t)g0 * batch_records) - 4;   // (the marks' "1 +")
        uint32_t carry = 0;
        // One step = 64 * ILP records: `fetch` finds every record's run and requests it, `process` adds the records to
        // the tile.  The record loads are issued from inline asm so that the NEXT step's requests can be in flight while
        // this step's records go through the palette and the tile (ACC_PIPE): the compiler would wait for them at once.
        // A step's results are its ILP records and one word with the ILP palette rows (one byte each).
        static_assert(ILP >= 2 && ILP <= 4, "up to four rows to a word");
        auto fetch = [&](const uint32_t v0, uint32_t (&rec)[ILP], uint32_t &rows) __attribute__((always_inline)) {
#if ACC_BYTE_MARKS
            // ONE exchange for the whole step: the runs that start inside it drop their lane number (+1) as a byte, every
            // lane reads its ILP positions at once, the ILP max-scans are independent instruction chains (the wait states
            // of one are filled by the others), and the marks themselves come from the runs' lanes by ds_bpermute —
            // two waits on the LDS pipe per step instead of ILP.  (carry: the run number + 1 here, the mark below)
            unsigned char *mk8 = reinterpret_cast<unsigned char *>(mk);
            const bool starts = c != 0u && excl - v0 < 64u * ILP;
            if (starts) mk8[excl - v0] = (unsigned char)(lane + 1u);
            wave_sync();
            uint32_t mm[ILP];
#pragma unroll
            for (int k = 0; k < ILP; ++k) mm[k] = mk8[k * 64 + lane];
            wave_sync();
            if (starts) mk8[excl - v0] = 0;
#pragma unroll
            for (int k = 0; k < ILP; ++k) mm[k] = wave_incl_maxscan(mm[k]);
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                mm[k] = max(mm[k], carry);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)mm[k], 63);
                mm[k] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((mm[k] - 1u) << 2), (int)mark_l);
            }
#endif
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                const uint32_t lo = v0 + k * 64, v = lo + lane;
#if ACC_BYTE_MARKS
                const uint32_t m = mm[k];
#else
                if (c != 0u && excl - lo < 64u) mk[excl - lo] = mark_l;
                wave_sync();                 // lanes exchange data through LDS: without it the compiler
                uint32_t m = mk[lane];       // forwards this lane's own earlier "= 0" into the load
                wave_sync();
                mk[lane] = 0u;
                m = wave_incl_maxscan(m);
                m = max(m, carry);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
#endif
                // byte k of `rows` = the mark's low byte
                rows = __builtin_amdgcn_perm(m, rows, k == 0 ? 0x03020104u : k == 1 ? 0x03020400u : k == 2 ? 0x03040100u : 0x04020100u);
#ifdef ACC_X_NOLOG       /* timing experiments only (tools/exp_accum_parts.sh): synthesised records */
                rec[k] = ((m * 2654435761u + v * 40503u) & ((1u << (TWL + FL_TILE_H_LOG2 + 8u)) - 1u));
#elif 0
                // 3-byte records: the two aligned words around the record's byte address in ONE 8-byte load (needs 4-byte
                // alignment only), the record cut out with v_alignbyte — an unaligned 4-byte load measured +31 %
                rec[k] = 0u;
                if (v < total) {
                    const size_t ba = ((size_t)g0 * batch_records + (v + (m >> 8) - 1u)) * 3u;
                    struct __attribute__((packed, aligned(4))) W2 { uint32_t lo, hi; };
                    const W2 w = *reinterpret_cast<const W2 *>(reinterpret_cast<const unsigned char *>(log) + (ba & ~(size_t)3));
                    rec[k] = __builtin_amdgcn_alignbyte(w.hi, w.lo, (uint32_t)ba & 3u) & 0xffffffu;
                }
#else
                // scalar base + 32-bit byte offset (at most 4 * 65 * batch_records); positions past the end of the
                // virtual array read the group's first record (and are not used)
                const uint32_t voff = v < total ? ((m >> 8) + v) << 2 : 4u;
                asm volatile("global_load_dword %0, %1, %2" ACC_LOAD_MOD : "=v"(rec[k]) : "v"(voff), "s"(gbase) : "memory");
#endif
            }
        };
        auto process = [&](const uint32_t v0, uint32_t (&rec)[ILP], const uint32_t rows) __attribute__((always_inline)) {
            bool live[ILP];
            u64 val[ILP];
            auto gather = [&](const int k) __attribute__((always_inline)) {
#ifdef ACC_X_NOPAL
                val[k] = (1ull << 54) | (rows & 0xffu) | (rec[k] & 0xffu);
#else
                val[k] = pal[__builtin_amdgcn_perm(rows, rec[k], 0x0c0c0000u | ((4u + k) << 8))];      // (row << 8) | colour byte: FL_PAL_W == 256
#endif
            };
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                live[k] = v0 + k * 64 + lane < total;
                if (k < ACC_GATHER_AHEAD) gather(k);
            }
            // A cell that takes most of the samples (a point attractor takes all of them) would receive
            // thousands of adds between the moment its count passes the drain threshold and the moment
            // the drain executes — enough to carry out of the 10-bit count.  When at least 48 lanes of the
            // step's first 64 records share one cell, the step is examined record set by record set:
            // the lanes that share the first lane's cell are summed in registers (at most 64 hits: no
            // field overflows) and go straight to the float accumulator (hot_group, out of line; the
            // test on the first set costs four instructions per 256 records).
            {
                const uint32_t o0 = rec[0] >> 8;
                if (__builtin_expect(__popcll(__ballot(live[0] && o0 == (uint32_t)__builtin_amdgcn_readfirstlane((int)o0))) >= 48, 0)) {
#pragma unroll
                    for (int k = ACC_GATHER_AHEAD; k < ILP; ++k) gather(k);          // (the rare path looks at every record's entry)
#pragma unroll
                    for (int k = 0; k < ILP; ++k) {
                        const uint32_t off = rec[k] >> 8;
                        const uint32_t off0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
                        const bool mine = live[k] && off == off0;
                        const unsigned long long grp = __ballot(mine);
                        if (__popcll(grp) >= 16) {
                            hot_group(mine ? val[k] : 0ull, grp, (ty * FL_TILE_H + (off0 >> TWL)) * astride + tx * TW + (off0 & (TW - 1u)), out4);
                            live[k] = live[k] && !mine;
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                const uint32_t off = rec[k] >> 8;                                // (ly << TWL) | lx
                if (k + ACC_GATHER_AHEAD < ILP) gather(k + ACC_GATHER_AHEAD);           // palette entries are requested ACC_GATHER_AHEAD records ahead of their add
                if (!live[k]) continue;
#ifdef ACC_X_NOATOM
                const u64 old = tile[off ^ 1u]; if (val[k] == 0x1234567ull) tile[off] = old;
#else
                const u64 old = __hip_atomic_fetch_add(tile + off, val[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
                if ((uint32_t)(old >> 32) >= (128u << 23)) {                     // 256 hits: drained early, three quarters of the count's range left for adds in flight
                    const u64 cur = __hip_atomic_exchange(tile + off, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if ((uint32_t)(cur >> 32) != 0u) {
                        const uint32_t px = tx * TW + (off & (TW - 1u)), py = ty * FL_TILE_H + (off >> TWL);
                        spill_cell(cur, py * astride + px, out4);
                    }
                }
            }
        };
        // the wait names the records it is for: nothing of `process` can be scheduled above it
#define ACC_WAIT(n, r) do { if constexpr (ILP == 4) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(r[0]), "+v"(r[1]), "+v"(r[ILP - 2]), "+v"(r[ILP - 1]) :: "memory"); \
                            else if constexpr (ILP == 3) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(r[0]), "+v"(r[1]), "+v"(r[ILP - 1]) :: "memory"); \
                            else asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(r[0]), "+v"(r[1]) :: "memory"); } while (0)
#define ACC_WAIT_NEWER(r) do { if constexpr (ILP == 4) ACC_WAIT(4, r); else if constexpr (ILP == 3) ACC_WAIT(3, r); else ACC_WAIT(2, r); } while (0)       /* all but the ILP newest loads */
        constexpr uint32_t STEP = 64 * ILP;
#if ACC_PIPE && !defined(ACC_X_NOLOG)
        // Two sets of results alternate (A, B): while one set's records are added, the other's are on their way.  Each
        // set has ONE place where it is requested, and B's processing trails into the next iteration: a second place
        // (a prologue, say) would make the compiler merge two definitions, i.e. copy registers whose loads are still
        // in flight (tools/check_asm_atomics.py looks for exactly that in the assembly).
        uint32_t recA[ILP] = {}, recB[ILP] = {}, rowsA = 0u, rowsB = 0u;
#ifdef ACC_X_TIMES
#define ACC_T(sum) do { const uint32_t t_ = (uint32_t)__builtin_amdgcn_s_memtime(); sum += t_ - x_t; x_t = t_; } while (0)
        uint32_t x_t = (uint32_t)__builtin_amdgcn_s_memtime();
#else
#define ACC_T(sum) do { } while (0)
#endif
        for (uint32_t v0 = 0; v0 < total; v0 += 2 * STEP) {
            ACC_T(x_other); fetch(v0, recA, rowsA); ACC_T(x_fetch);
            // (the marker tells tools/check_asm_atomics.py that the branch around this block is the first step's: nothing of B in flight)
            if (v0 != 0u) { asm volatile("; acc-guarded-block"); ACC_WAIT_NEWER(recB); ACC_T(x_wait); process(v0 - STEP, recB, rowsB); ACC_T(x_proc); }       // B is older than A: A stays in flight
            if (v0 + STEP < total) { fetch(v0 + STEP, recB, rowsB); ACC_T(x_fetch); ACC_WAIT_NEWER(recA); } else ACC_WAIT(0, recA);
            ACC_T(x_wait); process(v0, recA, rowsA); ACC_T(x_proc);
#ifdef ACC_X_TIMES
            x_steps += v0 + STEP < total ? 2 : 1;
#endif
        }
        if (((total + STEP - 1u) / STEP & 1u) == 0u && total != 0u) {                     // an even number of steps: the last B
            ACC_WAIT(0, recB);
            process((total - 1u) / STEP * STEP, recB, rowsB);
        }
#else
        for (uint32_t v0 = 0; v0 < total; v0 += STEP) {
            uint32_t rec[ILP], rows = 0u;
            fetch(v0, rec, rows);
#if !defined(ACC_X_NOLOG)
            ACC_WAIT(0, rec);
#endif
            process(v0, rec, rows);
        }
#endif
#undef ACC_WAIT
#undef ACC_T
#undef ACC_WAIT_NEWER
    }
    cb = ce;
    }
    __syncthreads();

    // add the tile to the global packed accumulator: one row segment of 64 cells per wave
    // instruction (coalesced atomics), draining cells that reach 512 hits
#ifdef ACC_X_TIMES
    if (tid == 0 && blockIdx.x < ACC_X_MAXWG) {
        unsigned long long *o = acc_wg_steps[blockIdx.x];
        o[0] = x_fetch; o[1] = x_wait; o[2] = x_proc; o[3] = (uint32_t)__builtin_amdgcn_s_memtime() - x_t0; o[4] = x_steps;
    }
    __syncthreads();
    if (tid == 0 && blockIdx.x < ACC_X_MAXWG) acc_wg_times[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime();
#endif
#ifndef ACC_NO_DRAIN     /* timing experiment only: tools/exp_drain.sh */
    // ACC_ADD_ILP returning atomics per thread are in flight before the first result is looked at (one at a time,
    // the loop was eight serial round trips to L2: 6.5 us of a 43 us workgroup)
    for (uint32_t i0 = tid; i0 < CELLS; i0 += blockDim.x * ACC_ADD_ILP) {
        u64 old[ACC_ADD_ILP];
        uint32_t big = 0;
#pragma unroll
        for (uint32_t k = 0; k < ACC_ADD_ILP; ++k) {
            const uint32_t i = i0 + k * blockDim.x;
            old[k] = 0ull;
            if (i < CELLS) {
                const u64 v = tile[i];
                const uint32_t px = tx * TW + (i & (TW - 1u)), py = ty * FL_TILE_H + (i >> TWL);
                if (v != 0ull && px < astride && py < aheight) {
                    // A packed add must never carry the 10-bit count past 1023.  Each cell receives at
                    // most `nparts` adds per launch (one per workgroup of its tile) onto a flushed cell,
                    // so chunks below big_thr = 1024/nparts hits are safe; larger ones go straight to the floats.
                    // (Round 4's FLAME_FLUSH_LAST — one flush per frame instead of one per launch, for the on-die log experiment —
                    // sized this threshold for a drain at 256 hits while cells drain at 512 (`full` below): removed in round 5.)
                    if ((uint32_t)(v >> 54) >= big_thr) big |= 1u << k;
                    else                                         // (as asm: the compiler waits for every returning atomic at the end of its branch)
                        asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0" : "+v"(old[k]) : "v"(atom + py * astride + px), "v"(v) : "memory");
                }
            }
        }
        static_assert(ACC_ADD_ILP == 8 || ACC_ADD_ILP == 4, "the wait below names every result");
        if constexpr (ACC_ADD_ILP == 8)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(old[4]), "+v"(old[5]), "+v"(old[6]), "+v"(old[7]) :: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]) :: "memory");
#pragma unroll
        for (uint32_t k = 0; k < ACC_ADD_ILP; ++k) {
            const bool full = (uint32_t)(old[k] >> 32) >= (256u << 23);
            if (full || ((big >> k) & 1u)) {                      // rare: the cell is looked up again
                const uint32_t i = i0 + k * blockDim.x;
                const uint32_t gi = (ty * FL_TILE_H + (i >> TWL)) * astride + tx * TW + (i & (TW - 1u));
                if (full) {
                    const u64 cur = __hip_atomic_exchange(atom + gi, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(cur >> 32) != 0u) spill_cell(cur, gi, out4);
                } else
                    spill_cell(tile[i], gi, out4);
            }
        }
    }
#endif
#ifdef ACC_X_TIMES
    __syncthreads();
    if (tid == 0 && blockIdx.x < ACC_X_MAXWG) acc_wg_times[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---- the accumulate of the packed log (FL_LOG_PACK3: 128x64 tiles, three 21-bit records per 64-bit word) ----------------------
// Same structure as k_accum_tiles — a tile per P workgroups, a wave per 64 directory entries, their runs walked as one virtual
// array — in units of log WORDS, and written for instruction count: by the SQ counters the accumulate is bound by vector-ALU
// issue (74-86 % of the kernel; profiles/r06_sq_counters_k_accum_tiles*.json), not by the bytes it streams.
//  * a step is 64 words (one per lane: a group of 64 runs of ~5 words fills its last step to 94 %, steps of 128 to 78 %), and
//    THREE steps rotate through three register sets — two in flight while one is added;
//  * only a run's first and last word can hold other tiles' records: the run's directory lane prepares both slot masks once,
//    a word takes them when it is the run's first (it found a mark at its own position) or last (its position is the run's end);
//  * a slot that is not this tile's still runs through the palette and the tile — with the address of a zero entry (LDS byte 0:
//    one v_bfe_i32 + one v_and per slot, no exec mask, no branch) — and adds nothing;
//  * the LDS layout is fixed at compile time and addressed by number, so that palette and tile bases are immediate offsets of the
//    LDS instructions, and palette column and cell offset go from the 64-bit word to their addresses in one shift + one mask each.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u64 lds_u64_t;
typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
__device__ __forceinline__ lds_u64_t *lds64(uint32_t byte) { return (lds_u64_t *)(uintptr_t)byte; }
__device__ __forceinline__ lds_u8_t *lds8(uint32_t byte) { return (lds_u8_t *)(uintptr_t)byte; }
// LDS of a workgroup, in bytes: [0, 16) a zero palette entry; the waves' marks (64 bytes each); the palette rows in use; the tile
constexpr uint32_t P3_MARKS = 64u, P3_PAL = 2048u, P3_TILE = P3_PAL + ACC_ROWS_MAX * FL_PAL_W * 8u, P3_LDS = P3_TILE + FL_TILE_CELLS * 8u;
static_assert(P3_MARKS + (ACC_THREADS / 64) * 64 <= P3_PAL && P3_PAL == FL_PAL_W * 8u && ACC_ROWS_MAX < 15, "marks below the rows; row + 1 is a nibble");
__global__ void __launch_bounds__(ACC_THREADS, 8) __attribute__((amdgpu_num_sgpr(80)))
k_accum_tiles_p3(const uint32_t *__restrict__ log, const uint32_t *__restrict__ dir,
                 const u64 *__restrict__ palette, u64 *__restrict__ atom, float *__restrict__ out4,
                 uint32_t tiles_x, uint32_t nparts, uint32_t nbatch_total, uint32_t batch_words,
                 uint32_t nslots, uint32_t astride, uint32_t aheight, uint32_t rows_cap, uint32_t big_thr, uint32_t gang, uint32_t nbins)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t TWL = 7u, TW = 1u << TWL, CELLS = TW * FL_TILE_H;
    static_assert(FL_PAL_W == 256 && FL_REC_BITS == 21, "8-bit palette column, 13-bit cell offset");
    if ((uint32_t)(uintptr_t)(lds_u8_t *)smem != 0u) __builtin_trap();            // the dynamic LDS is all the LDS of this kernel: it starts at byte 0
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwaves = blockDim.x >> 6;
    uint32_t bin, part;                                            // workgroup -> (tile, part), see k_accum_tiles
    if (gang == 0u) { bin = blockIdx.x / nparts; part = blockIdx.x % nparts; }
    else {
        const uint32_t loc = blockIdx.x >> 3, gq = (loc / gang) * 8u + (blockIdx.x & 7u);
        bin = (gq / nparts) * gang + loc % gang; part = gq % nparts;
        if (bin >= nbins) return;
    }
    const uint32_t tx = bin % tiles_x, ty = bin / tiles_x;
    const uint32_t b_lo = (uint32_t)((u64)nbatch_total * part / nparts);
    const uint32_t b_hi = (uint32_t)((u64)nbatch_total * (part + 1) / nparts);
    const uint32_t *drow = dir + (size_t)bin * nbatch_total;
    const uint32_t per_slot = nbatch_total / nslots;
    const uint32_t spr = nslots / FL_PAL_H;                       // slots per palette row
    const float inv_spr = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (float)spr)));
    const float inv_ps = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (float)per_slot)));
    const uint32_t chunk_slots = (rows_cap - 1u) * spr;
    if (b_lo >= b_hi) return;                                      // no batches for this part (tiny launches): nothing to add
    const uint32_t mk8 = P3_MARKS + wv * 64u;                      // this wave's marks
    const uint32_t lane8 = (lane + 1u) << 3;
    u32x2_t wA = {}, wB = {}, wC = {};                             // the three register sets of the record loop: a log word each ...
    uint32_t mA = 0u, mB = 0u, mC = 0u;                            // ... and (its palette row + 1) << 6 | slots of this tile

    for (uint32_t cb = b_lo; cb < b_hi;) {
    const uint32_t cs_lo = cb / per_slot;
    const uint32_t ce = min(b_hi, (cs_lo + chunk_slots) * per_slot);
    const uint32_t row_lo = cs_lo / spr, nrows = min(rows_cap, (uint32_t)FL_PAL_H - row_lo);
    u64 stagev[3];                                                 // rows_cap * 256 <= 3 * blockDim.x
    uint32_t tid_c = tid;                                          // (opaque: see k_accum_tiles)
    asm volatile("" : "+v"(tid_c));
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const uint32_t i = tid_c + q * blockDim.x;
        stagev[q] = i < nrows * FL_PAL_W ? palette[row_lo * FL_PAL_W + i] : 0ull;
    }
    uint32_t e_next = cb + wv * 64 + lane < ce ? drow[cb + wv * 64 + lane] : 0u;      // the first group's directory words
    if (cb == b_lo) {
        for (uint32_t i = tid; i < CELLS; i += blockDim.x) *lds64(P3_TILE + i * 8u) = 0ull;
        *lds8(mk8 + lane) = 0;
        if (tid < 2u) *lds64(tid * 8u) = 0ull;
    }
    __syncthreads();                                               // readers of the previous chunk's rows are done
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const uint32_t i = tid_c + q * blockDim.x;
        if (i < nrows * FL_PAL_W) *lds64(P3_PAL + i * 8u) = stagev[q];
    }
    __syncthreads();
    for (uint32_t g0v = cb + wv * 64; g0v < ce; g0v += nwaves * 64) {
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g0v);      // wave-uniform
        const uint32_t s0 = g0 / per_slot, rem0 = g0 - s0 * per_slot;
        const uint32_t batch = g0 + lane;
        const uint32_t e = e_next;
        e_next = batch + nwaves * 64 < ce ? drow[batch + nwaves * 64] : 0u;
        // This lane's run: records rfirst .. rlast of its batch, i.e. words fw .. lw of the batch's region; the first word's slots
        // below rfirst % 3 and the last word's above rlast % 3 belong to the neighbouring tiles.  (n / 3 = n * 43691 >> 17 below 98304.)
        const uint32_t cnt = e & 0xffffu, rfirst = e >> 16, rlast = rfirst + cnt - 1u;
        const uint32_t fw = __umul24(rfirst, 43691u) >> 17, lw = __umul24(rlast, 43691u) >> 17;
        const uint32_t c = cnt ? lw - fw + 1u : 0u;                                // words of the run
        const uint32_t lo_mask = (7u << (rfirst - 3u * fw)) & 7u, hi_mask = 7u >> (2u - (rlast - 3u * lw));
        const uint32_t incl = wave_incl_scan_b(c, lane);
        const uint32_t excl = incl - c;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        // what a word needs from its run, fetched from the run's lane with ds_bpermute (the marks that are exchanged are lane numbers):
        //   w1 = 1 + the word's index in the log counted from the group's first batch - its position v in the virtual array
        //   w2 = (position of the run's last word) << 10 | (palette row among the staged rows + 1) << 6 | slot mask of the last word << 3 | of the first
        const uint32_t rslot_l = s0 + (uint32_t)(((float)(rem0 + lane) + 0.5f) * inv_ps);          // exact: small integers
        const uint32_t row_l = (uint32_t)(((float)rslot_l + 0.5f) * inv_spr) - row_lo;
        const uint32_t w1_l = lane * batch_words + fw - excl + 1u;
        const uint32_t w2_l = ((incl - 1u) << 10) | ((row_l + 1u) << 6) | (hi_mask << 3) | lo_mask;
        const unsigned char *gbase = reinterpret_cast<const unsigned char *>(log) + (size_t)g0 * batch_words * 8u - 8u;      // (w1's "1 +")
        uint32_t carry = 0;
        // fetch: the run of word v0 + lane, its load, and one register of what the adds need
        auto fetch = [&](const uint32_t v0, u32x2_t &w, uint32_t &meta) __attribute__((always_inline)) {
            uint32_t voff = 8u;                                                    // steps past the end read the group's first word
            meta = 0u;
            if (v0 < total) {
                const bool starts = cnt != 0u && excl - v0 < 64u;
                if (starts) *lds8(mk8 + excl - v0) = (unsigned char)(lane + 1u);
                wave_sync();
                const uint32_t raw = *lds8(mk8 + lane);
                wave_sync();
                if (starts) *lds8(mk8 + excl - v0) = 0;
                uint32_t mm = max(wave_incl_maxscan(raw), carry);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)mm, 63);
                const int src = (int)(mm << 2) - 4;                                 // byte address of the run's lane
                const uint32_t m1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)w1_l);
                const uint32_t m2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)w2_l);
                const uint32_t v = v0 + lane;
                const bool inside = v < total;
                // the run's first word takes the first-word mask (and whatever lies above it: the last-word mask is three clean bits)
                const uint32_t lm = raw != 0u ? m2 : 7u, hm = (m2 >> 10) == v ? __builtin_amdgcn_ubfe(m2, 3, 3) : 7u;
                meta = (m2 & 0x3c0u) | (inside ? lm & hm : 0u);
                // (positions past the end take a word of their own from the group's first batch: the SAME word in all of them would send
                // their adds of nothing to the same three cells, one lane at a time)
                voff = inside ? (m1 + v) << 3 : lane8;
            }
            // the ONE place this set is requested (the steps behind the last one still issue a load, so that every wait below is "all but the two newest")
            asm volatile("global_load_dwordx2 %0, %1, %2" ACC_LOAD_MOD : "=v"(w) : "v"(voff), "s"(gbase) : "memory");
        };
        auto process = [&](const u32x2_t w, const uint32_t meta) __attribute__((always_inline)) {
            const uint32_t wl = w.x, wh = w.y;
            const uint32_t rowbase = (meta & 0x3c0u) << 5;                          // P3_PAL + row * 2048
            // slot j: palette column = bits 21 j .. + 8, cell = bits 21 j + 8 .. + 13 of the word; both as byte offsets (x 8)
            const uint32_t col8[3] = {(wl << 3) & 0x7f8u, (wl >> 18) & 0x7f8u, (wh >> 7) & 0x7f8u};
            const uint32_t cell8[3] = {(wl >> 5) & 0xfff8u, __builtin_amdgcn_alignbit(wh, wl, 26) & 0xfff8u, (wh >> 15) & 0xfff8u};
            u64 val[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint32_t live = (uint32_t)__builtin_amdgcn_sbfe((int)meta, j, 1);          // all ones, or zero: a slot of another tile reads the zero entry
                val[j] = *lds64((col8[j] | rowbase) & live);
            }
            // A cell that takes most of the samples: when at least 48 lanes' first slots share lane 0's cell, the word's three
            // record sets are examined and the lanes that share a cell go to the float accumulator in one piece (see k_accum_tiles).
            if (__builtin_expect((uint32_t)__popcll(__ballot(cell8[0] == (uint32_t)__builtin_amdgcn_readfirstlane((int)cell8[0]))) >= 48u, 0)) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const bool live = (meta >> j & 1u) != 0u;
                    const unsigned long long lv = __ballot(live);
                    if (lv == 0ull) continue;
                    const uint32_t off0 = (uint32_t)__builtin_amdgcn_readlane((int)cell8[j], (int)__builtin_ctzll(lv)) >> 3;
                    const bool mine = live && cell8[j] >> 3 == off0;
                    const unsigned long long grp = __ballot(mine);
                    if (__popcll(grp) >= 16) {
                        hot_group_sw(mine ? val[j] : 0ull, grp, (ty * FL_TILE_H + (off0 >> TWL)) * astride + tx * TW + (off0 & (TW - 1u)), out4);
                        if (mine) val[j] = 0ull;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                lds_u64_t *cp = lds64(P3_TILE + cell8[j]);
                const u64 old = __hip_atomic_fetch_add(cp, val[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // 256 hits: drained early, three quarters of the count's range left for adds in flight (each add's result is looked
                // at before the lane's next add is issued)
                if (__builtin_expect(__ballot((uint32_t)(old >> 32) >= (128u << 23)) != 0ull, 0)) {
                    if ((uint32_t)(old >> 32) >= (128u << 23)) {
                        const u64 cur = __hip_atomic_exchange(cp, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if ((uint32_t)(cur >> 32) != 0u) {
                            const uint32_t off = cell8[j] >> 3, px = tx * TW + (off & (TW - 1u)), py = ty * FL_TILE_H + (off >> TWL);
                            spill_cell(cur, py * astride + px, out4);
                        }
                    }
                }
            }
        };
        // Steps s = 0, 1, ... of 64 words; step s is requested into set s % 3 and added once the two steps behind it have been requested,
        // so that two steps' loads are in flight while a third is added.  Each set has ONE place where it is requested and one where it
        // is added (a second definition would make the compiler copy registers whose loads are still in flight:
        // tools/check_asm_atomics.py), hence the loop in threes; every wait is "all but the two newest loads" and names its set.
        const uint32_t nsteps = (total + 63u) >> 6;
#define P3_WAIT(w) asm volatile("s_waitcnt vmcnt(2)" : "+v"(w) :: "memory")
        for (uint32_t s = 0; s < nsteps + 2u; s += 3u) {              // (no early exits: a fetch behind the last step is a branch and one load)
            // (every wait is executed, whether or not its set is then added — at the ends of the array it is satisfied at once —, so that no
            // path reaches a set's next request with an earlier load of the set still in flight: its registers are the lookup's scratch)
            fetch(s << 6, wA, mA);
            P3_WAIT(wB); if (s >= 2u && s - 2u < nsteps) process(wB, mB);
            fetch((s + 1u) << 6, wB, mB);
            P3_WAIT(wC); if (s >= 1u && s - 1u < nsteps) process(wC, mC);
            fetch((s + 2u) << 6, wC, mC);
            P3_WAIT(wA); if (s < nsteps) process(wA, mA);
        }
        // (the loads behind the last step: nothing may be in flight into a set when the next group's first fetch uses the
        // set's registers — to the compiler they are free until the load defines them)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(wA), "+v"(wB), "+v"(wC) :: "memory");
#undef P3_WAIT
    }
    cb = ce;
    }
    __syncthreads();
    add_tile_to_cells<TWL>(reinterpret_cast<const u64 *>(smem + P3_TILE), atom, out4, tx, ty, astride, aheight, big_thr);
}

void launch_accum_tiles(hipStream_t st, const uint32_t *log, const uint32_t *dir, const u64 *palette,
                        u64 *atom, float *out4, uint32_t tiles_x, uint32_t nbins, uint32_t nparts,
                        uint32_t nbatch_total, uint32_t batch_records, uint32_t nslots,
                        uint32_t astride, uint32_t aheight, bool wide)
{
    // k_flush has emptied every packed cell before this launch, and a cell receives at most one add per workgroup of its tile:
    // chunks below 1024 / nparts hits cannot carry the 10-bit count past 1023 (larger ones go straight to the floats)
    const uint32_t big_thr = 1024u / nparts;
    // LDS: the tile, 64 marks per wave, and as many palette rows (2 KB each) as the slot range of one
    // workgroup touches — capped by what lets two narrow workgroups (one wide) share a CU's 160 KB
    const uint32_t spr = nslots / FL_PAL_H, slots_per_part = (nslots + nparts - 1) / nparts;
    if (batch_records > 65536u) abort();        // directory words hold 16-bit counts; the kernel's marks 24 bits of 64 * batch_records
    const uint32_t want = (slots_per_part + spr - 1) / spr + 1;
    // Gangs of 32 adjacent tiles for images of more than 512 tiles (round 5, profiles/r05_bin_gang.txt).  A batch's records are sorted
    // by tile, so the runs of adjacent tiles share cache lines — at 4K / 8K a run is 8-14 records, a quarter to a half of a line — and
    // with the parts of ONE tile as consecutive workgroups the tiles that share a line passed it at different times, each on whatever
    // XCD its workgroup had landed on: k_accum_tiles fetched 14.5 GB per 8K launch for 4.3 GB of records (3.8 / 1.1 at 4K).  Ganged:
    // 7.0 GB (1.8), 2504 -> 2304 us per launch at 8K, 627 -> 531 at 4K.  At 1080p (288 tiles, runs of 14 records, 16 parts) the gangs
    // change nothing (347 -> 355 us): off.  FLAME_BIN_GANG forces a gang size (0: off).
    static const int gang_env = getenv("FLAME_BIN_GANG") ? atoi(getenv("FLAME_BIN_GANG")) : -1;
    const uint32_t gang = gang_env >= 0 ? (uint32_t)gang_env : (nbins > 512u ? 32u : 0u);
    // gangs = ceil(nbins / gang) * nparts, dealt to the XCDs eight at a time
    const uint32_t ngangs = gang ? ((nbins + gang - 1) / gang) * nparts : 0u;
    const uint32_t grid = gang ? ((ngangs + 7u) / 8u) * 8u * gang : nbins * nparts;
    if (wide) {
        const uint32_t rows = want < 2u ? 2u : want > 12u ? 12u : want;
        static unsigned long long attr = 0;
        ensure_max_dynamic_lds((const void *)k_accum_tiles<FL_TILE_W_WIDE_LOG2>, attr);
        hipLaunchKernelGGL((k_accum_tiles<FL_TILE_W_WIDE_LOG2>), dim3(grid), dim3(1024),
                           (FL_TILE_H << FL_TILE_W_WIDE_LOG2) * 8 + 1024 * 4 + rows * FL_PAL_W * 8, st,
                           log, dir, palette, atom, out4, tiles_x, nparts, nbatch_total, batch_records, nslots, astride, aheight, rows, big_thr, gang, nbins);
        return;
    }
    const uint32_t rows = want < 2u ? 2u : want > (uint32_t)ACC_ROWS_MAX ? (uint32_t)ACC_ROWS_MAX : want;
    static unsigned long long attr = 0;
#if FL_LOG_PACK3
    {   // 128x64 tiles: three records per 64-bit log word, a batch's region in such words (flame_device.h)
        static unsigned long long attr3 = 0;
        ensure_max_dynamic_lds((const void *)k_accum_tiles_p3, attr3);
        hipLaunchKernelGGL(k_accum_tiles_p3, dim3(grid), dim3(ACC_THREADS), P3_LDS, st,
                           log, dir, palette, atom, out4, tiles_x, nparts, nbatch_total, fl_pack3_words(batch_records), nslots, astride, aheight, rows, big_thr, gang, nbins);
        return;
    }
#endif
    ensure_max_dynamic_lds((const void *)k_accum_tiles<7u>, attr);
    hipLaunchKernelGGL((k_accum_tiles<7u>), dim3(grid), dim3(ACC_THREADS), FL_TILE_CELLS * 8 + ACC_THREADS * 4 + rows * FL_PAL_W * 8, st,
                       log, dir, palette, atom, out4, tiles_x, nparts, nbatch_total, batch_records, nslots, astride, aheight, rows, big_thr, gang, nbins);
}
