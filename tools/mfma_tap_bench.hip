// mfma_tap_bench.hip — can the DE's tap loop give its bilinear parts to the matrix pipe?  (round 5)
//
// The tap loop of de.hip is bound by vector-ALU issue (DESIGN.md 4.3 "Round 4").  Two parts of a tap are bilinear in
// (centre c, tap position q): the colour term of the exponent, n_q . C'_c + y_q * Dl_c, and the accumulation
// sum_q f(c, q) * (w_q n_q, w_q).  With lanes grouped four to a block — four centres in a line along the filter direction —
// v_mfma_f32_4x4x1_16b_f32 computes, per block, a 4 x 4 patch of (q, c) pairs from one register per operand: it runs on the
// matrix pipe, beside the vector ALU (f32 MFMA has the same FLOP rate as v_fma_f32, MI355X_MICROARCH.md, so the gain is
// the second pipe, not a faster one).  This benchmark times one GROUP of that loop (4 tap positions x 64 centres):
//   5 ds_read_b128 (own q's vector, 4 x |ds|w^dpow, 4 x plane term, 4 x spatial coefficient, 4 x channel of w n)
//   4 MFMA colour term (K = 4, chained on one accumulator, C initialised from the plane-term read)
//   1 MFMA density difference x_c - x_q (C = x_c in four registers)
//   4 x { e = t - |d| ; exp2 ; * spatial ; wsum += }         = 16 vector instructions, 4 of them v_exp_f32
//   4 MFMA accumulate (A = f_i, B = channel of tap i, K = 1 each)
// against the loop as shipped (two taps per step: 26 VALU + 2 v_exp_f32 + 4 ds_read_b128, vgpr_bank_bench.hip's mix but
// written in C++ like the variants here), all with 8 waves per SIMD and 64 registers.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_tap_bench tools/mfma_tap_bench.hip && tools/mfma_tap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
#define GROUPS 9            /* groups of four tap positions per centre (36 positions for 31 taps) */
#define REPS 256

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// MODE 0: MFMA form, LDS reads; 1: MFMA form without the LDS reads; 2: MFMA form without the vector part (MFMA + reads only)
// 3: shipped form (VALU only), LDS reads; 4: shipped form without LDS reads; 5: MFMA form, vector part only (no MFMA)
template <int MODE>
__global__ void __launch_bounds__(256, 8) k(float *out, const float *in, int reps)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) lds[i] = in[i & 255] * 1e-3f;
    __syncthreads();
    const f4 *L = reinterpret_cast<const f4 *>(lds);
    // per-centre constants
    const float Cx = in[lane], Cy = in[lane + 1], Cz = in[lane + 2], Dl = in[lane + 3], xc = in[lane + 4];
    f4 xc4 = {xc, xc, xc, xc};
    f4 acc = {0, 0, 0, 0};
    float wsum = 0.0f;
    int base = (tid >> 2) * 5;          // block-uniform read position (broadcast reads), lane part for the own-q read
    int own = tid;
    if (MODE <= 2 || MODE == 5) {
        for (int rep = 0; rep < reps; ++rep) {
            const f4 *Lo = L + own, *Lb = L + base, *Ls = L + ((own >> 1) + lane), *Lv = L + base + (lane & 3) * 64;
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                f4 qv, x4, bz4, sp4, v4;
                if (MODE == 0 || MODE == 2) {
                    qv = Lo[g * 7]; x4 = Lb[g * 11]; bz4 = Lb[g * 11 + 1];
                    sp4 = Ls[g * 13]; v4 = Lv[g * 11 + 2];
                } else {
                    qv = acc * 0.5f + (float)g; x4 = xc4 * (float)(g + 1); bz4 = xc4 + (float)g; sp4 = xc4; v4 = xc4 - (float)g;
                    asm volatile("" : "+v"(qv), "+v"(x4), "+v"(bz4), "+v"(sp4), "+v"(v4));
                }
                f4 t = bz4, d = xc4;
                if (MODE != 5) {
                    t = __builtin_amdgcn_mfma_f32_4x4x1f32(qv.x, Cx, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_4x4x1f32(qv.y, Cy, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_4x4x1f32(qv.z, Cz, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_4x4x1f32(qv.w, Dl, t, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_4x4x1f32(x4.x, -1.0f, d, 0, 0, 0);
                } else { t = t + qv; d = d - x4; }
                f4 f;
                if (MODE != 2) {
                    f.x = ex2(t.x - fabsf(d.x)) * sp4.x; f.y = ex2(t.y - fabsf(d.y)) * sp4.y;
                    f.z = ex2(t.z - fabsf(d.z)) * sp4.z; f.w = ex2(t.w - fabsf(d.w)) * sp4.w;
                    wsum += f.x; wsum += f.y; wsum += f.z; wsum += f.w;
                } else f = t + d;
                if (MODE != 5) {
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(f.x, v4.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(f.y, v4.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(f.z, v4.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(f.w, v4.w, acc, 0, 0, 0);
                } else acc += f * v4;
            }
            own = (own + 17) & 511; base = (base + 3) & 255;
        }
    } else {
        float cds = xc;
        for (int rep = 0; rep < reps; ++rep) {
            const f4 *Lo = L + own;
#pragma unroll
            for (int g = 0; g < GROUPS * 4; ++g) {       // one tap each
                f4 a, b;
                if (MODE == 3) { a = Lo[g * 7]; b = Lo[g * 11 + 1]; }
                else { a = acc * 0.5f + (float)g; b = xc4 + (float)g; asm volatile("" : "+v"(a), "+v"(b)); }
                float t = fmaf(b.y, Dl, b.z);
                t = fmaf(a.z, Cz, t); t = fmaf(a.y, Cy, t); t = fmaf(a.x, Cx, t);
                const float dd = cds - b.x;
                const float e = t - fabsf(dd);
                float f = ex2(e) * in[g & 15];
                const float fw = f * a.w;
                wsum += f;
                acc.x = fmaf(fw, a.x, acc.x); acc.y = fmaf(fw, a.y, acc.y); acc.z = fmaf(fw, a.z, acc.z); acc.w += fw;
            }
            own = (own + 17) & 511;
        }
    }
    const float s = acc.x + acc.y + acc.z + acc.w + wsum;
    if (s == 1.2345e-33f) out[tid] = s;
}

template <int MODE> static void run(const char *name, float *d, float *in, int ncu)
{
    const int wps = 8, blocks = ncu * wps;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 16384, 0, d, in, REPS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 16384, 0, d, in, REPS);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 8 waves, each REPS * GROUPS groups of four tap positions
    const double ns_group = ms / 5 * 1e6 / ((double)REPS * GROUPS * wps);
    printf("%-72s %7.2f ns per group of 4 tap positions per wave (SIMD time), %6.2f ns per tap\n", name, ns_group, ns_group / 4);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    float *d, *in; hipMalloc(&d, 4096); hipMalloc(&in, 4096);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (float)(i % 97);
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    const int ncu = p.multiProcessorCount;
    run<3>("shipped form: 13 VALU + 1 v_exp_f32 per tap, 2 ds_read_b128 per tap", d, in, ncu);
    run<4>("shipped form without the LDS reads", d, in, ncu);
    run<0>("MFMA form: 9 MFMA 4x4x1 + 16 VALU (4 exp) + 5 ds_read_b128 per 4 taps", d, in, ncu);
    run<1>("MFMA form without the LDS reads", d, in, ncu);
    run<2>("MFMA form, matrix pipe + reads only (no exp / spatial / wsum)", d, in, ncu);
    run<5>("MFMA form, vector part only (MFMAs replaced by one VALU op each side)", d, in, ncu);
    return 0;
}
