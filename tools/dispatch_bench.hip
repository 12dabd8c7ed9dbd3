// dispatch_bench.hip — what launching a direction's workgroups costs by itself (round 5): an (almost) empty kernel in the DE's
// launch shapes — N workgroups of 256 threads with the DE's LDS request and register count — timed over many launches.
//   hipcc --offload-arch=gfx950 -O2 -o tools/dispatch_bench tools/dispatch_bench.hip && tools/dispatch_bench
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WORK>
__global__ void __launch_bounds__(256, 8) k(float *out, int n)
{
    extern __shared__ float lds[];
    float acc = 0.0f;
    // keep ~64 registers live so that the wave's register allocation is the DE's
    float r[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) r[i] = (float)(threadIdx.x + i);
    if (WORK) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 48; ++i) r[i] = fmaf(r[i], 1.0001f, 0.5f);
        }
    }
#pragma unroll
    for (int i = 0; i < 48; ++i) acc += r[i];
    if (threadIdx.x == 0) lds[0] = acc;
    __syncthreads();
    if (acc == 1.2345e-33f) out[blockIdx.x] = lds[0];
}
template <int WORK> static void run(const char *name, float *d, int nwg, int lds, int n)
{
    hipFuncSetAttribute((const void *)k<WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<WORK>, dim3(nwg), dim3(256), lds, 0, d, n);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<WORK>, dim3(nwg), dim3(256), lds, 0, d, n);
    (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %6d workgroups, %5d B LDS: %7.2f us per launch = %6.1f ns per workgroup\n", name, nwg, lds, ms / reps * 1e3, ms / reps * 1e6 / nwg);
}
int main()
{
    float *d; (void)hipMalloc(&d, 1 << 20);
    run<0>("empty body (register init + one barrier)", d, 8540, 18752, 0);
    run<0>("empty body", d, 8540, 1024, 0);
    run<0>("empty body", d, 2048, 18752, 0);
    run<0>("empty body", d, 34160, 18752, 0);
    run<1>("48 dependent-free FMAs x 100 per thread (~6.5 us alone)", d, 8540, 18752, 100);
    run<1>("the same, 2048 workgroups (one per slot)", d, 2048, 18752, 100);
    run<1>("the same, 4270 workgroups of twice the work", d, 4270, 18752, 200);
    return 0;
}
