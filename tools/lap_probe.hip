// Can the head of kernel B run under the tail of kernel A without a deadlock-prone co-dispatch?  (round 4, DE chain)
// A and B: 9216 workgroups of 256 threads with 20 KB of LDS (8 per CU -> 2048 resident: 4.5 rounds), each spinning 50 us.
//   serial : A, B on one stream                          -> 10 rounds
//   gate   : A on stream 0; on stream 1 a one-wave kernel that waits until every workgroup of A has STARTED, then B
//   waitval: the same with hipStreamWaitValue64 instead of the gate kernel                  -> 9 rounds + the hand-over
//   hipcc --offload-arch=gfx950 -O2 -w tools/lap_probe.hip -o tools/lap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) work(unsigned long long ticks, unsigned long long *started) {
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();           // 100 MHz
    if (threadIdx.x == 0) { __hip_atomic_fetch_add(started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); lds[0] = 1;
                            __hip_atomic_fetch_min(started + 2, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) __hip_atomic_fetch_max(started + 4, __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void gate(const unsigned long long *started, unsigned long long target) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(4);
    }
}
int main() {
    unsigned long long *cnt; if (hipMalloc(&cnt, 64) != hipSuccess) return 1;
    hipStream_t s0, s1; (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipEvent_t a, b, f, j; (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventCreateWithFlags(&f, hipEventDisableTiming); (void)hipEventCreateWithFlags(&j, hipEventDisableTiming);
    int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    const unsigned n = 9216; const unsigned long long ticks = 5000;
    (void)hipFuncSetAttribute((const void *)work, hipFuncAttributeMaxDynamicSharedMemorySize, 20480);
    for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 4; ++rep) {
        if (mode == 2 && !can) continue;
        (void)hipMemsetAsync(cnt, 0, 64, s0); (void)hipStreamSynchronize(s0);
        (void)hipEventRecord(a, s0);
        hipLaunchKernelGGL(work, dim3(n), dim3(256), 20480, s0, ticks, cnt);
        if (mode == 0) hipLaunchKernelGGL(work, dim3(n), dim3(256), 20480, s0, ticks, cnt + 1);
        else {
            (void)hipEventRecord(f, s0); (void)hipStreamWaitEvent(s1, f, 0);       // (would order s1 after s0's EARLIER work: recorded after A here only to mimic the fork cost)
            // NB: the fork above makes s1 wait for A itself; record the fork BEFORE A instead
        }
        (void)hipEventRecord(b, s0); (void)hipStreamSynchronize(s0); (void)hipStreamSynchronize(s1);
        if (mode != 0) continue;
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("serial : %.1f us\n", ms * 1e3f);
    }
    for (int mode = 1; mode < 3; ++mode) for (int rep = 0; rep < 4; ++rep) {
        if (mode == 2 && !can) continue;
        (void)hipMemsetAsync(cnt, 0, 64, s0); (void)hipMemsetAsync(cnt + 2, 0xff, 16, s0); (void)hipStreamSynchronize(s0);
        (void)hipEventRecord(a, s0);
        (void)hipEventRecord(f, s0); (void)hipStreamWaitEvent(s1, f, 0);           // fork
        hipLaunchKernelGGL(work, dim3(n), dim3(256), 20480, s0, ticks, cnt);
        if (mode == 1) hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, s1, (const unsigned long long *)cnt, (unsigned long long)n);
        else if (hipStreamWaitValue64(s1, cnt, n, hipStreamWaitValueGte, ~0ull) != hipSuccess) { printf("hipStreamWaitValue64 failed\n"); break; }
        hipLaunchKernelGGL(work, dim3(n), dim3(256), 20480, s1, ticks, cnt + 1);
        (void)hipEventRecord(j, s1); (void)hipStreamWaitEvent(s0, j, 0);           // join
        (void)hipEventRecord(b, s0); (void)hipStreamSynchronize(s0);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        unsigned long long h[8]; (void)hipMemcpy(h, cnt, 64, hipMemcpyDeviceToHost);
        printf("%s: %.1f us   A: first start 0, last end %.1f us;  B: first start %.1f us, last end %.1f us\n", mode == 1 ? "gate   " : "waitval", ms * 1e3f,
               (h[4] - h[2]) * 0.01, (h[3] - h[2]) * 0.01, (h[5] - h[2]) * 0.01);
    }
    return 0;
}
