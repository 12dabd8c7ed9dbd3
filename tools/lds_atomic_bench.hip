// lds_atomic_bench.hip — LDS throughput of the accumulate's two scattered operations, per CU, wall clock (HIP events):
//   ds_add_rtn_u64 at random cells of a 64 KB tile, and ds_read_b64 gathers from a 10 KB palette,
// with 2 x 16 waves per CU as k_accum_tiles runs them (one result looked at per add, or four in flight).
//   hipcc --offload-arch=gfx950 -O2 -o tools/lds_atomic_bench tools/lds_atomic_bench.hip && tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;

template <int MODE> __global__ void __launch_bounds__(1024, 8) k(u64 *out, int iters, uint32_t spread)
{
    extern __shared__ u64 lds[];                       // 8192 cells + 1280 palette entries
    for (uint32_t i = threadIdx.x; i < 8192 + 1280; i += blockDim.x) lds[i] = i;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    u64 acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { x = x * 1664525u + 1013904223u; a[k] = (x >> 8); }
        if (MODE == 0) {               // four serial returning adds (each result is looked at before the next add)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u64 o = __hip_atomic_fetch_add(lds + (a[k] & spread), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((uint32_t)(o >> 32) >= 0x40000000u) acc += o;
            }
        } else if (MODE == 1) {        // four returning adds in flight
            u64 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __hip_atomic_fetch_add(lds + (a[k] & spread), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int k = 0; k < 4; ++k) if ((uint32_t)(o[k] >> 32) >= 0x40000000u) acc += o[k];
        } else if (MODE == 2) {        // four palette gathers (8 bytes at a random one of 1280 entries)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += lds[8192 + a[k] % 1280u];
        } else if (MODE == 3) {        // adds without return
#pragma unroll
            for (int k = 0; k < 4; ++k) __hip_atomic_fetch_add(lds + (a[k] & spread), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {                       // the random numbers alone
            acc += a[0] ^ a[1] ^ a[2] ^ a[3];
        }
    }
    if (acc == 0x1234567u) out[0] = acc;
}

template <int MODE> static double run(const char *what, int ncu, uint32_t spread)
{
    u64 *d; hipMalloc(&d, 64);
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 2000, nb = ncu * 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(1024), (8192 + 1280) * 8, 0, d, iters, spread);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(1024), (8192 + 1280) * 8, 0, d, iters, spread);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_ops_per_cu = 32.0 * iters * 4;            // 32 waves per CU, 4 operations per iteration
    const double ns = ms * 1e6 / wave_ops_per_cu;
    printf("%-64s %7.3f ms: %6.2f ns per wave operation per CU\n", what, ms, ns);
    hipFree(d);
    return ns;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    const double base = run<4>("random numbers only (loop overhead)", ncu, 8191);
    run<0>("ds_add_rtn_u64, random cell of 8192, one at a time", ncu, 8191);
    run<1>("ds_add_rtn_u64, random cell of 8192, four in flight", ncu, 8191);
    run<3>("ds_add_u64 (no return), random cell of 8192", ncu, 8191);
    run<0>("ds_add_rtn_u64, random cell of 256 (hot region), one at a time", ncu, 255);
    run<2>("ds_read_b64 gather, random entry of 1280", ncu, 0);
    printf("(loop overhead %.2f ns is included in every line; k_accum_tiles<7> at 412 us per 2^28 records spends %.1f ns per 64 records per CU)\n",
           base, 412e3 / (268435456.0 / 64 / ncu));
    return 0;
}
