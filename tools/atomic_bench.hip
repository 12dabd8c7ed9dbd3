// Micro-benchmark: 64-bit packed atomic scatter rate on MI355X, by scope, footprint,
// spatial distribution and cell layout.  Informs the accumulate design of flame_iter
// (DESIGN.md "atomic ceiling").  Standalone: hipcc --offload-arch=gfx950 -O3 atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t xcc_id() {
    // HW_REG_XCC_ID = 20, bits [3:0]
    return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7;
}

__device__ __forceinline__ uint32_t mwc(uint32_t &s, uint32_t &c, uint32_t a) {
    u64 t = (u64)a * s + c; s = (uint32_t)t; c = (uint32_t)(t >> 32); return s;
}

enum { M_AGENT = 0, M_WG_XCD = 1, M_AGENT_RTN = 2, M_STORE = 3, M_AGENT_U32 = 4, M_AGENT_F32 = 5, M_SYSTEM = 6 };
enum { D_UNIFORM = 0, D_IFS = 1, D_HOT = 2, D_COAL = 3, D_WINDOW = 4, D_PAIRS = 5 };

// width/height in cells; layout 0 = row-major, 1 = 4x4 tiles (one 128-B line per tile)
template <int MODE, int DIST, int LAYOUT>
__global__ void __launch_bounds__(256) k_scatter(u64 *hist, size_t copy_stride, int width, int height,
                                                 int iters, u64 *sink)
{
    uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = gt * 2654435761u + 12345u, c = gt ^ 0x9e3779b9u, a = 4294967118u - 2 * (gt & 1023);
    float x = 0.3f, y = 0.3f;
    u64 acc = 0;
    u64 *base = hist;
    if (MODE == M_WG_XCD) base = hist + (size_t)xcc_id() * copy_stride;
    for (int i = 0; i < iters; ++i) {
        uint32_t r = mwc(s, c, a);
        uint32_t ix, iy;
        if (DIST == D_UNIFORM) {
            ix = (uint32_t)(((u64)r * (uint32_t)width) >> 32);
            iy = (uint32_t)(((u64)mwc(s, c, a) * (uint32_t)height) >> 32);
        } else if (DIST == D_COAL || DIST == D_WINDOW || DIST == D_PAIRS) {
            // wave-level patterns: a random base per wave (lane 0's draw), then
            //   COAL:   lane l -> base + l            (64 consecutive cells = 4 lines)
            //   WINDOW: lane l -> base + random(512)  (same 4 KB window)
            //   PAIRS:  lane l -> base + (l/2)*977    (two lanes per cell, cells scattered)
            uint32_t ru = __builtin_amdgcn_readfirstlane(r);
            uint32_t cells = (uint32_t)width * (uint32_t)height;
            uint32_t base_i = (uint32_t)(((u64)ru * (cells - 65536u)) >> 32);
            uint32_t lane = threadIdx.x & 63;
            uint32_t off = DIST == D_COAL ? lane : (DIST == D_WINDOW ? (mwc(s, c, a) & 511u) : (lane >> 1) * 977u);
            uint32_t lin = base_i + off;
            ix = lin % (uint32_t)width; iy = lin / (uint32_t)width;
        } else {
            // wave-uniform map choice like the flame kernel: use lane-0's r
            uint32_t ru = __builtin_amdgcn_readfirstlane(r);
            uint32_t k = ru % 3u;
            float vx = k == 0 ? 0.0f : (k == 1 ? 1.0f : 0.5f);
            float vy = k == 2 ? 1.0f : 0.0f;
            x = 0.5f * (x + vx); y = 0.5f * (y + vy);
            if (DIST == D_HOT && (r & 0xff) < 16) { x = 0.5f; y = 0.5f; }   // ~6% into one pixel
            // decorrelate lanes a little
            x += ((int)(r >> 8) & 0xffff) * (1.0f / 65536.0f / 4096.0f);
            ix = (uint32_t)(x * (width - 1)); iy = (uint32_t)(y * (height - 1));
            if (ix >= (uint32_t)width) ix = width - 1;
            if (iy >= (uint32_t)height) iy = height - 1;
        }
        size_t idx;
        if (LAYOUT == 0) idx = (size_t)iy * width + ix;
        else idx = ((size_t)(iy >> 2) * (width >> 2) + (ix >> 2)) * 16 + ((iy & 3) << 2) + (ix & 3);
        u64 val = (1ull << 54) | ((u64)(r & 0xff) << 36) | ((r >> 8) & 0xff) << 18 | ((r >> 16) & 0xff);
        if (MODE == M_AGENT) __hip_atomic_fetch_add(base + idx, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == M_SYSTEM) __hip_atomic_fetch_add(base + idx, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (MODE == M_WG_XCD) __hip_atomic_fetch_add(base + idx, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == M_AGENT_RTN) acc += __hip_atomic_fetch_add(base + idx, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == M_STORE) base[idx] = val;
        else if (MODE == M_AGENT_U32) __hip_atomic_fetch_add((uint32_t *)(base + idx), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == M_AGENT_F32) __hip_atomic_fetch_add((float *)(base + idx), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 0x1234567 || s == 0xdeadbeef) sink[0] = acc + s;
}

__global__ void k_copy(float4 *dst, const float4 *src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) dst[i] = src[i];
}

__global__ void k_xcc_census(uint32_t *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

template <int MODE, int DIST, int LAYOUT>
static void run(const char *name, u64 *hist, size_t copy_stride, int ncopies, int w, int h, u64 *sink, std::vector<u64> &hbuf)
{
    const int blocks = 256 * 8 * 2, iters = 512;
    size_t cells = (size_t)w * h;
    CK(hipMemset(hist, 0, copy_stride * ncopies * sizeof(u64)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_scatter<MODE, DIST, LAYOUT><<<blocks, 256>>>(hist, copy_stride, w, h, 16, sink);   // warm
    CK(hipDeviceSynchronize());
    CK(hipMemset(hist, 0, copy_stride * ncopies * sizeof(u64)));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_scatter<MODE, DIST, LAYOUT><<<blocks, 256>>>(hist, copy_stride, w, h, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double n = (double)blocks * 256 * iters;
    // verify conservation for packed-u64 modes
    const char *ok = "-";
    if (MODE == M_AGENT || MODE == M_WG_XCD || MODE == M_AGENT_RTN || MODE == M_SYSTEM) {
        hbuf.resize(copy_stride * ncopies);
        CK(hipMemcpy(hbuf.data(), hist, hbuf.size() * sizeof(u64), hipMemcpyDeviceToHost));
        // count field = bits 54.. ; sum of low fields may overflow into it, so count via full u64 sum / compare modulo: use a wide sum
        unsigned __int128 tot = 0;
        for (size_t i = 0; i < hbuf.size(); ++i) tot += hbuf[i];
        // expected: recompute is expensive; use count of adds: each adds (1<<54)+low; we check only (tot >> 54) >= n and low part small
        // low parts: 3 fields of <=255 at bits 0,18,36 -> per add < 2^44, so total count = floor(tot / 2^54) exact when n*2^44 < 2^54 * ... not exact; report ratio
        double cnt = (double)(tot >> 54);
        ok = (cnt >= n && cnt < n * 1.002) ? "conserved" : "LOST";
        (void)cells;
    }
    printf("%-44s %4dx%-4d copies=%d  %8.3f ms  %8.2f Gatom/s  %s\n", name, w, h, ncopies, ms, n / ms * 1e-6, ok);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s  CUs=%d  clock=%d MHz  L2=%d\n", p.name, p.multiProcessorCount, p.clockRate / 1000, p.l2CacheSize);
    // census
    uint32_t *dc; CK(hipMalloc(&dc, 64 * 4));
    k_xcc_census<<<64, 64>>>(dc);
    uint32_t hc[64]; CK(hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost));
    printf("xcc of blocks 0..31:"); for (int i = 0; i < 32; ++i) printf(" %u", hc[i]); printf("\n");

    // HBM copy bandwidth
    {
        size_t n = (size_t)64 << 20;   // 64M float4 = 1 GiB
        float4 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
        CK(hipMemset(a, 1, n * 16));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            k_copy<<<2048 * 4, 256>>>(b, a, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("float4 copy 1 GiB: %.3f ms  %.1f GB/s (r+w)\n", ms, 2.0 * n * 16 / ms * 1e-6);
        }
        CK(hipFree(a)); CK(hipFree(b));
    }

    struct Fp { int w, h; } fps[] = { {1952, 1104}, {3872, 2192}, {7712, 4352} };
    size_t maxcells = (size_t)7712 * 4352;
    u64 *hist, *sink; CK(hipMalloc(&hist, maxcells * 8 * sizeof(u64))); CK(hipMalloc(&sink, 64));
    std::vector<u64> hbuf;
    for (auto fp : fps) {
        int w = fp.w, h = fp.h; size_t cs = (size_t)w * h;
        printf("--- footprint %dx%d (%.1f MB per copy)\n", w, h, cs * 8 / 1e6);
        run<M_AGENT,     D_UNIFORM, 0>("agent  nortn uniform rowmajor", hist, cs, 1, w, h, sink, hbuf);
        run<M_WG_XCD,    D_UNIFORM, 0>("wg/xcd nortn uniform rowmajor", hist, cs, 8, w, h, sink, hbuf);
        run<M_AGENT_RTN, D_UNIFORM, 0>("agent  rtn   uniform rowmajor", hist, cs, 1, w, h, sink, hbuf);
        run<M_STORE,     D_UNIFORM, 0>("plain store  uniform rowmajor", hist, cs, 1, w, h, sink, hbuf);
        run<M_AGENT_U32, D_UNIFORM, 0>("agent u32    uniform rowmajor", hist, cs, 1, w, h, sink, hbuf);
        run<M_AGENT_F32, D_UNIFORM, 0>("agent f32    uniform rowmajor", hist, cs, 1, w, h, sink, hbuf);
        run<M_SYSTEM,    D_UNIFORM, 0>("system nortn uniform rowmajor", hist, cs, 1, w, h, sink, hbuf);
        run<M_AGENT,     D_COAL, 0>("agent  nortn wave-coalesced 64 cells", hist, cs, 1, w, h, sink, hbuf);
        run<M_AGENT,     D_WINDOW, 0>("agent  nortn wave in 4KB window", hist, cs, 1, w, h, sink, hbuf);
        run<M_AGENT,     D_PAIRS, 0>("agent  nortn lane pairs same cell", hist, cs, 1, w, h, sink, hbuf);
        run<M_AGENT_U32, D_COAL, 0>("agent u32 wave-coalesced", hist, cs, 1, w, h, sink, hbuf);
        run<M_AGENT_F32, D_COAL, 0>("agent f32 wave-coalesced", hist, cs, 1, w, h, sink, hbuf);
        run<M_STORE,     D_COAL, 0>("plain store wave-coalesced", hist, cs, 1, w, h, sink, hbuf);
        run<M_AGENT,     D_IFS, 0>("agent  nortn ifs     rowmajor", hist, cs, 1, w, h, sink, hbuf);
        run<M_AGENT,     D_IFS, 1>("agent  nortn ifs     tiled4x4", hist, cs, 1, w, h, sink, hbuf);
        run<M_WG_XCD,    D_IFS, 0>("wg/xcd nortn ifs     rowmajor", hist, cs, 8, w, h, sink, hbuf);
        run<M_WG_XCD,    D_IFS, 1>("wg/xcd nortn ifs     tiled4x4", hist, cs, 8, w, h, sink, hbuf);
        run<M_AGENT,     D_HOT, 0>("agent  nortn ifs+hot rowmajor", hist, cs, 1, w, h, sink, hbuf);
        run<M_WG_XCD,    D_HOT, 0>("wg/xcd nortn ifs+hot rowmajor", hist, cs, 8, w, h, sink, hbuf);
    }
    return 0;
}
