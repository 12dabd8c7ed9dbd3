// tools/copy_bench.hip — which float4 copy kernel streams fastest on this box (round 6: the denominator of bench.py's DE fraction).
// hipcc --offload-arch=gfx950 -O3 tools/copy_bench.hip -o tools/copy_bench && tools/copy_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NT_LD, int NT_ST, int UNROLL>
__global__ void __launch_bounds__(256) k_copy(f4 *__restrict__ dst, const f4 *__restrict__ src, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256u;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (size_t)UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) if (i + k * stride < n4) v[k] = NT_LD ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) if (i + k * stride < n4) { if (NT_ST) __builtin_nontemporal_store(v[k], dst + i + k * stride); else dst[i + k * stride] = v[k]; }
    }
}
template <int NT_LD, int NT_ST, int UNROLL>
static void run(const char *name, f4 *b, const f4 *a, size_t n4, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((k_copy<NT_LD, NT_ST, UNROLL>), dim3(blocks), dim3(256), 0, 0, b, a, n4);
    hipEventRecord(e0, 0);
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((k_copy<NT_LD, NT_ST, UNROLL>), dim3(blocks), dim3(256), 0, 0, b, a, n4);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks %6d  %7.1f GB/s\n", name, blocks, 2.0 * n4 * 16 / (ms / 10 * 1e-3) / 1e9);
}
int main()
{
    const size_t n4 = (size_t)1 << 26;          // 1 GiB each way
    f4 *a, *b; hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16); hipMemset(a, 1, n4 * 16);
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536, 262144}) {
        run<0, 0, 1>("plain x1", b, a, n4, blocks);
        run<0, 0, 4>("plain x4", b, a, n4, blocks);
        run<0, 1, 4>("nt store x4", b, a, n4, blocks);
        run<1, 1, 4>("nt load + store x4", b, a, n4, blocks);
        run<1, 1, 8>("nt load + store x8", b, a, n4, blocks);
        run<0, 0, 8>("plain x8", b, a, n4, blocks);
    }
    hipMemcpyDtoDAsync(b, a, n4 * 16, 0); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0); for (int k = 0; k < 10; ++k) hipMemcpyDtoDAsync(b, a, n4 * 16, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpyDtoD                              %7.1f GB/s\n", 2.0 * n4 * 16 / (ms / 10 * 1e-3) / 1e9);
    return 0;
}
