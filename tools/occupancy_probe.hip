// occupancy_probe.hip — how many workgroups of a given size and LDS footprint does a CU of this GPU really hold?
// (a) what hipOccupancyMaxActiveBlocksPerMultiprocessor says; (b) measured: every workgroup of a grid of
// 4 x CUs workgroups reports when it started and ended (s_memrealtime) after spinning ~20 us: the number of
// workgroups whose lifetimes overlap the first one's is the real residency.
//   hipcc --offload-arch=gfx950 -O2 -o tools/occupancy_probe tools/occupancy_probe.hip && tools/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int NT>
__global__ void __launch_bounds__(NT) k_probe(unsigned long long *out, int spin)
{
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) lds[0] = 1;
    __syncthreads();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() + lds[0]; }
}

// the same with a register footprint: the clobber lists make the compiler allocate up to the named registers
#define PROBE_R(name, VTOP, STOP)                                                                                     \
    __global__ void __launch_bounds__(1024) name(unsigned long long *out, int spin)                                   \
    {                                                                                                                 \
        extern __shared__ unsigned char lds[];                                                                        \
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                                               \
        asm volatile("" ::: VTOP, STOP);                                                                             \
        if (threadIdx.x == 0) lds[0] = 1;                                                                             \
        __syncthreads();                                                                                              \
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);         \
        __syncthreads();                                                                                              \
        if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() + lds[0]; } \
    }
PROBE_R(k_probe_v56, "v55", "s8")
PROBE_R(k_probe_s80, "v8", "s79")
PROBE_R(k_probe_s88, "v8", "s87")
PROBE_R(k_probe_s95, "v8", "s95")
PROBE_R(k_probe_s48, "v8", "s47")
PROBE_R(k_probe_s64, "v8", "s63")
PROBE_R(k_probe_s72, "v8", "s71")
PROBE_R(k_probe_s74, "v8", "s73")
PROBE_R(k_probe_s76, "v8", "s75")
PROBE_R(k_probe_v56_s80, "v55", "s79")
PROBE_R(k_probe_v64_s80, "v63", "s79")

template <class K> static void probe_k(K kern, const char *what, int NT, size_t ldsb, int ncu)
{
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int api = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, kern, NT, ldsb);
    const int nb = ncu * 4 * (1024 / NT);
    unsigned long long *d;
    hipMalloc(&d, nb * 16);
    hipMemset(d, 0, nb * 16);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(NT), ldsb, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    unsigned long long first_end = ~0ull;
    for (int i = 0; i < nb; ++i) first_end = std::min(first_end, h[2 * i + 1]);
    int resident = 0;
    for (int i = 0; i < nb; ++i) resident += h[2 * i] < first_end;
    printf("%-18s %4d threads, %6zu B of LDS: API says %d per CU; measured %d resident at once = %.2f per CU\n", what, NT, ldsb, api, resident, (double)resident / ncu);
    hipFree(d);
}

template <int NT> static void probe(size_t ldsb, int ncu)
{
    hipFuncSetAttribute((const void *)k_probe<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int api = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k_probe<NT>, NT, ldsb);
    const int nb = ncu * 4 * (1024 / NT);
    unsigned long long *d;
    hipMalloc(&d, nb * 16);
    hipMemset(d, 0, nb * 16);
    hipLaunchKernelGGL(k_probe<NT>, dim3(nb), dim3(NT), ldsb, 0, d, 2000);     // 20 us at 100 MHz
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    unsigned long long first_end = ~0ull;
    for (int i = 0; i < nb; ++i) first_end = std::min(first_end, h[2 * i + 1]);
    int resident = 0;
    for (int i = 0; i < nb; ++i) resident += h[2 * i] < first_end;            // started before the first one ended
    printf("%4d threads, %6zu B of LDS: API says %d per CU; measured %d workgroups resident at once = %.2f per CU\n", NT, ldsb, api, resident, (double)resident / ncu);
    hipFree(d);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs, %zu B of LDS per workgroup max, %d B per CU\n", p.gcnArchName, ncu, p.sharedMemPerBlock, (int)p.maxSharedMemoryPerMultiProcessor);
    for (size_t l : {65536ul, 73728ul, 79872ul, 81920ul, 40960ul}) probe<1024>(l, ncu);
    for (size_t l : {36864ul, 40960ul, 20480ul}) probe<512>(l, ncu);
    for (size_t l : {18432ul, 20480ul, 26624ul}) probe<256>(l, ncu);
    probe_k(k_probe_v56, "56 VGPR", 1024, 79872, ncu);
    probe_k(k_probe_s48, "48 SGPR", 1024, 79872, ncu);
    probe_k(k_probe_s64, "64 SGPR", 1024, 79872, ncu);
    probe_k(k_probe_s72, "72 SGPR", 1024, 79872, ncu);
    probe_k(k_probe_s74, "74 SGPR", 1024, 79872, ncu);
    probe_k(k_probe_s76, "76 SGPR", 1024, 79872, ncu);
    probe_k(k_probe_s80, "80 SGPR", 1024, 79872, ncu);
    probe_k(k_probe_s88, "88 SGPR", 1024, 79872, ncu);
    probe_k(k_probe_s95, "96 SGPR", 1024, 79872, ncu);
    probe_k(k_probe_v56_s80, "56 VGPR + 80 SGPR", 1024, 79872, ncu);
    probe_k(k_probe_v64_s80, "64 VGPR + 80 SGPR", 1024, 79872, ncu);
    probe_k(k_probe_v56_s80, "56 VGPR + 80 SGPR", 1024, 8192, ncu);
    // smaller workgroups with 80 / 96 SGPRs: how many WAVES per SIMD fit
    probe_k(k_probe_s80, "80 SGPR", 256, 1024, ncu);
    probe_k(k_probe_s95, "96 SGPR", 256, 1024, ncu);
    probe_k(k_probe_s64, "64 SGPR", 256, 1024, ncu);
    return 0;
}
