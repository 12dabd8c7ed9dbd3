// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h says "not supported on GFX9xx".)
// Two launches of 64 workgroups that each spin for 200 us: ~200 us in all = they overlap, ~400 us = they do not.
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/anyorder_probe && tools/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long *out) {
    const unsigned long long t0 = __builtin_readcyclecounter();           // s_memtime: 100 MHz on this part
    unsigned long long t = t0;
    while (t - t0 < ticks) { __builtin_amdgcn_s_sleep(8); t = __builtin_readcyclecounter(); }
    if (threadIdx.x == 0) out[blockIdx.x] = t - t0;
}
int main() {
    unsigned long long *d; hipMalloc(&d, 1 << 16);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    // calibrate the counter: one launch of N ticks
    for (int flags = 0; flags < 2; ++flags) for (int rep = 0; rep < 3; ++rep) {
        const unsigned long long ticks = 400000;
        hipEventRecord(a, s);
        for (int k = 0; k < 2; ++k) {
            void *args[] = {(void *)&ticks, (void *)&d};
            hipError_t e = hipExtLaunchKernel((const void *)spin, dim3(64), dim3(64), args, 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0);
            if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
        }
        hipEventRecord(b, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("flags=%d  two launches of %llu ticks each: %.1f us\n", flags, ticks, ms * 1e3f);
    }
    return 0;
}
