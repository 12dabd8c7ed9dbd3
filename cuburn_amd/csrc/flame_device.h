// flame_device.h — device-side helpers shared by the gfx950 kernels: MWC RNG, packed
// accumulator cell, fast math wrappers.  (Product code; independent of oracle/.)
#pragma once
#ifdef __HIPCC_RTC__
// hipRTC: the HIP device declarations are built in; no system headers
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned long long uint64_t;
typedef long long int64_t;
typedef unsigned long size_t;
#include "flame_hip.h"
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/flame_hip.h"
#endif

typedef unsigned long long u64;

// Float constants as spelled by the reference's device prelude (cuburn/code/util.py:148-160)
#define FM_PI 3.14159274101257f
#define FM_PI_2 1.57079637050629f
#define FM_1_PI 0.31830987334251f
#define FM_2_PI 0.63661974668503f
#define FM_LOG2E 1.44269502162933f
#define FM_SQRT2 1.41421353816986f
#define FL_INV255 0.003921568859368562698f

// ---- multiply-with-carry RNG, cuburn/code/mwc.py:56-77 -----------------------------------
struct mwc_t { uint32_t mul, state, carry; };

__device__ __forceinline__ uint32_t mwc_next(mwc_t &s) {
    u64 t = (u64)s.mul * s.state + s.carry;       // v_mad_u64_u32
    s.state = (uint32_t)t;
    s.carry = (uint32_t)(t >> 32);
    return s.state;
}
// u32 -> f32 round-to-nearest, times 2^-32 (may return exactly 1.0f, like the reference)
__device__ __forceinline__ float mwc_next_01(mwc_t &s) { return (float)mwc_next(s) * (1.0f / 4294967296.0f); }
__device__ __forceinline__ float mwc_next_11(mwc_t &s) { return (float)(int32_t)mwc_next(s) * (1.0f / 2147483648.0f); }

// ---- packed 64-bit accumulator cell, include/flame_hip.h (3) -------------------------------
__device__ __forceinline__ void unpack_cell(u64 cell, float &y, float &u, float &v, float &d) {
    uint32_t hi = (uint32_t)(cell >> 32), lo = (uint32_t)cell;
    d = (float)(hi >> 22);
    y = (float)__builtin_amdgcn_ubfe(hi, 4, 18);
    u = (float)(((hi & 0xfu) << 14) | (lo >> 18));
    v = (float)(lo & 0x3ffffu);
}
// hot flag (0..3) -> sample weight 1, 2, 8, 32  (cuburn/code/iter.py:326)
__device__ __forceinline__ float hot_mult(uint32_t flag) { return flag ? (float)((1u << (flag << 1)) >> 1) : 1.0f; }

// ---- fast math: single hardware instructions (v_exp_f32, v_log_f32, v_rcp_f32, v_sqrt_f32,
// v_sin_f32, v_cos_f32), the counterpart of what nvcc -use_fast_math gives the reference
// (cuburn/code/util.py:96).  HIP's __powf/__expf/__fdividef expand to long IEEE-careful
// sequences (a __powf is ~80 instructions, __fdividef a full div_scale/div_fmas/div_fixup);
// these helpers do not: ~1 ulp, no denormal results, x/0 = inf, pow(0,y>0) = 0.
__device__ __forceinline__ float frcp(float a) { return __builtin_amdgcn_rcpf(a); }
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float flog2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * FM_LOG2E); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.69314718246460f; }
__device__ __forceinline__ float fpow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
// density^dpow of the DE filter (cuburn/code/filters.py:204,233).  powf(0, 0) is 1; the reference's fast-math
// powf (exp2f(y * __log2f(x))) and fpow give 0 * -inf = NaN there, i.e. a density power of exactly 0 breaks the
// reference wherever a pixel is empty.  Here the exponent 0 (kernel-uniform) means what it says: w^0 = 1.
__device__ __forceinline__ float de_pow(float w, float dpow) { return dpow == 0.0f ? 1.0f : fpow(w, dpow); }
__device__ __forceinline__ float fsqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
// v_sin/v_cos take revolutions; v_fract keeps the argument in the instruction's valid domain
__device__ __forceinline__ float fsin(float x) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915494309189532f)); }
__device__ __forceinline__ float fcos(float x) { return __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(x * 0.15915494309189532f)); }
__device__ __forceinline__ float ftan(float x) { return fsin(x) * frcp(fcos(x)); }

// cvt.rni.s32.f32 (cuburn/code/util.py:194-200): round-to-nearest-even, saturating, NaN -> 0
__device__ __forceinline__ uint32_t trunca(float f) {
    float c = fminf(fmaxf(f, -2147483648.0f), 2147483520.0f);      // NaN -> -2^31 here ...
    int32_t i = (int32_t)__builtin_rintf(c);
    if (f >= 2147483648.0f) i = 0x7fffffff;
    return (f != f) ? 0u : (uint32_t)i;                              // ... and 0 here
}

// ---- binned accumulate geometry: 128 x 64 pixel tiles; record = {bin 11 | ly 6 | lx 7 | ci 8} --------
#define FL_TILE_W 128u
#ifndef FL_TILE_H_LOG2
#define FL_TILE_H_LOG2 6u
#endif
#define FL_TILE_H (1u << FL_TILE_H_LOG2)
#define FL_TILE_CELLS (FL_TILE_W * FL_TILE_H)
#define FL_REC_BITS (15u + FL_TILE_H_LOG2)   /* ly + lx 7 + ci 8 */
// The sample log of 128x64 tiles holds THREE 21-bit records per aligned 64-bit word (record i of a sorted batch: bits 21 * (i % 3) ...
// of word i / 3; bit 63 unused): 2.67 bytes per sample instead of 4.  A tile's run may begin and end inside a word; the words at its
// ends are then shared with the neighbouring tiles' runs, and every reader takes the slots its directory entry names.
// 256x64 tiles (22-bit records) keep one record per 32-bit word.  -DFL_LOG_PACK3=0: 32-bit words everywhere (the format of rounds 1-5).
#ifndef FL_LOG_PACK3
#define FL_LOG_PACK3 1
#endif
// 64-bit words of one batch's region of the packed log (even: the regions are 16-byte aligned)
__host__ __device__ static inline uint32_t fl_pack3_words(uint32_t batch_records) { return ((batch_records + 2u) / 3u + 1u) & ~1u; }
#define FL_MAX_BINS 2047u                      /* 128x64 tiles: tile number shares the 32-bit staged record */
#define FL_TILE_W_WIDE_LOG2 8u                 /* 256x64 tiles for larger images (tile number staged separately) */
#define FL_MAX_BINS_WIDE 8191u
#ifndef FL_BIN_R_MAX
#define FL_BIN_R_MAX 16          /* rounds per sorted batch (records a thread holds in registers) */
#endif

// XCD id of the executing workgroup (HW_REG_XCC_ID, bits [3:0])
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 7; }
