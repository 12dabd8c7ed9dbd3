// variations.h — the flame variation functions, device side (gfx950).
//
// One wave applies ONE xform per round (wave-coherent selection), so the switch below is
// a scalar branch: no lane divergence, and the record pointers `v` / `xf` are wave-uniform
// (their loads become s_load).  Formulas: the flam3 variation set as used by
// cuburn/code/variations.py:22-988 (flam3 numbering, cuburn/genome/variations.py:28-127).
// v[0] = weight, v[1..] = genome parameters in sorted-name order then precalculated values
// (include/flame_hip.h (5)); xf = owning xform record (pre affine xx,xy,xo,yx,yy,yo).
#pragma once
#include "flame_device.h"

#define VW (w)
#define VP(i) (v[(i)])
#define OUT(a, b) do { ox += (a); oy += (b); } while (0)

// atan2(a, b) in ~24 instructions (the device library's atan2f is 38: it divides through frexp / ldexp so that the quotient
// survives operands near the ends of the float range, and sorts out zeros and infinities one compare at a time; the reference's is
// CUDA's library function under -use_fast_math, 2-3 ulp).  min / max -> one hardware reciprocal -> atan(q) = q + q^3 P(q^2) on [0, 1]
// (own least-squares minimax fit, degree 6 in q^2: 1.2e-7 absolute, 2.9e-7 relative in float arithmetic) -> octant fix-ups.
// atan2(0, 0) = 0; NaN in, NaN out; both operands infinite gives NaN (such a point is discarded by the caller either way).
// Ranges the lean forms do NOT cover (the tests' CPU restatement takes libm's answer there; such points lie outside any picture, and a walker
// that reaches them is re-seeded within a few rounds): v_atan2 with max(|a|, |b|) > 2^126 — the reciprocal is a denormal, flushed: the
// angle comes out as 0 or pi/2 — or with BOTH operands denormal (the FLT_MIN clamp); v_fmod with |b| > 2^126 (one subtraction
// where up to three may be needed); v_fmod_pi next to multiples of pi, where the result may be a few ulp of pi below 0 or at pi.
// -DFL_LIBM_MATH (through FLAME_RTC_FLAGS; FL_LIBM_ATAN2 is its round-4 name) compiles the device library's functions back in.
#if defined(FL_LIBM_ATAN2) && !defined(FL_LIBM_MATH)
#define FL_LIBM_MATH 1
#endif
#ifndef FL_LIBM_MATH
__device__ __forceinline__ float v_atan2(float a, float b)
{
    const float ax = fabsf(b), ay = fabsf(a);
    const float mx = fmaxf(fmaxf(ax, ay), 1.17549435e-38f), mn = fminf(ax, ay);
    const float q = mn * frcp(mx), s = q * q;
    float p = fmaf(s, -0.004355378448963165f, 0.023040037602186203f);
    p = fmaf(s, p, -0.05777344852685928f);
    p = fmaf(s, p, 0.09794224053621292f);
    p = fmaf(s, p, -0.13976578414440155f);
    p = fmaf(s, p, 0.19962702691555023f);
    p = fmaf(s, p, -0.3333165943622589f);
    float r = fmaf(q * s, p, q);
    r = ay > ax ? FM_PI_2 - r : r;
    r = b < 0.0f ? FM_PI - r : r;
    r = (a != a || b != b) ? __builtin_nanf("") : r;
    return copysignf(r, a);
}
#else
__device__ __forceinline__ float v_atan2(float a, float b) { return atan2f(a, b); }
#endif
// sinh / cosh in 5-16 instructions (the device library's are 120 and 116 — an extended-precision exponential each; the reference's are
// CUDA's library functions over ex2.approx under -use_fast_math).  h = e^|x| / 2 from ONE v_exp_f32 (the halving in the exponent, so that
// nothing overflows before the result does), cosh = h + 1/(4h), sinh = h - 1/(4h); below |x| = 0.5, where that difference cancels, the
// odd series to x^9 / 9! (next term 1.2e-11 relative).  The thirteen trigonometric / hyperbolic variations (sin .. coth) call both.
#ifndef FL_LIBM_MATH
__device__ __forceinline__ float v_cosh(float x)
{
    const float h = fexp2(fmaf(fabsf(x), FM_LOG2E, -1.0f));
    return fmaf(0.25f, frcp(h), h);
}
__device__ __forceinline__ float v_sinh(float x)
{
    const float ax = fabsf(x), h = fexp2(fmaf(ax, FM_LOG2E, -1.0f));
    const float big = fmaf(-0.25f, frcp(h), h), x2 = x * x;
    float p = fmaf(x2, 2.7557319e-6f, 1.9841270e-4f);
    p = fmaf(x2, p, 8.3333333e-3f);
    p = fmaf(x2, p, 1.6666667e-1f);
    return ax < 0.5f ? fmaf(x * x2, p, x) : copysignf(big, x);
}
#else
__device__ __forceinline__ float v_cosh(float x) { return coshf(x); }
__device__ __forceinline__ float v_sinh(float x) { return sinhf(x); }
#endif
// fmodf(a, b) in 13 instructions where the quotient fits a float's integers (|a / b| < 2^21): q = trunc(a / b) from the hardware reciprocal,
// r = a - q b in one fma, then the two ways q can be off by one put right (|r| >= |b|: q too small; r on the wrong side of 0: q too
// big) — the result has the sign of a, magnitude below |b|, and differs from the exact remainder by the fma's rounding only.
// Larger quotients (and b = 0, NaN) take the device library's exact loop, 45 instructions + 1-2 rounds, which every call paid before.
#ifndef FL_LIBM_MATH
__device__ __forceinline__ float v_fmod(float a, float b)
{
    const float qf = a * frcp(b);
    if (!(fabsf(qf) < 2097152.0f)) return fmodf(a, b);      // (the quotient from v_rcp_f32 is good to ~2 ulp: off by one at most)
    float r = fmaf(-truncf(qf), b, a);
    const float sb = copysignf(b, a);                       // |b| with a's sign
    r = fabsf(r) >= fabsf(b) ? r - sb : r;
    r = (r != 0.0f && (r < 0.0f) != (a < 0.0f)) ? r + sb : r;
    return copysignf(r, a);
}
#else
__device__ __forceinline__ float v_fmod(float a, float b) { return fmodf(a, b); }
#endif
// fmodf(a, pi) for a > 0 of a few pi (bipolar's wrap of an angle that left [-pi/2, pi/2]): a - pi floor(a / pi), 4 instructions where
// the device library's exact fmodf is a ~60-instruction loop that every lane of the wave sits through once one lane needs it
__device__ __forceinline__ float v_fmod_pi(float a) { return fmaf(-FM_PI, floorf(a * 0.318309886183791f), a); }
// box-muller style radius used by gaussian_blur / radial_blur (variations.py:318-323)
__device__ __forceinline__ float v_gauss_r(float w, mwc_t &r) {
    return w * 0.57736f * fsqrt(fdiv(-2.0f * flog2(mwc_next_01(r)), FM_LOG2E));
}

// The 16 cheapest / most common variations (flam3 numbers 0..15 minus the two that need
// invariants hoisted) are dispatched inline; everything else goes through ONE out-of-line
// function.  Keeping the big switch out of the iteration loop matters on AMDGPU: LLVM hoists
// loop-invariant pieces of *every* case (e.g. the fmod set-up of `rings`) into the loop
// preheader, where they would run on every round whatever the variation.
struct VarIO { float tx, ty, ox, oy; uint32_t state, carry; };

// (V: `const float *` at the variation's parameters, or a view that takes the words a kernel holds in registers from there: iter.hip VTail)
template <class V>
__device__ __forceinline__ bool apply_variation_body(int id, float w, const V v,
                                                     const float *__restrict__ xf,
                                                     float &tx, float &ty, float &ox, float &oy, mwc_t &r)
{
    const float r2 = fmaf(tx, tx, ty * ty);
    switch (id) {
    case 0: OUT(tx * VW, ty * VW); break;                                              // linear
    case 1: OUT(VW * fsin(tx), VW * fsin(ty)); break;                                  // sinusoidal
    case 2: { float k = fdiv(VW, r2); OUT(tx * k, ty * k); } break;                    // spherical
    case 3: { float s = fsin(r2), c = fcos(r2);                                        // swirl
              OUT(VW * (s * tx - c * ty), VW * (c * tx + s * ty)); } break;
    case 4: { float k = fdiv(VW, fsqrt(r2));                                           // horseshoe
              OUT(k * (tx - ty) * (tx + ty), 2.0f * tx * ty * k); } break;
    case 5: OUT(VW * v_atan2(tx, ty) * FM_1_PI, VW * (fsqrt(r2) - 1.0f)); break;       // polar
    case 6: { float a = v_atan2(tx, ty), rr = fsqrt(r2);                               // handkerchief
              OUT(VW * rr * fsin(a + rr), VW * rr * fcos(a - rr)); } break;
    case 7: { float sq = fsqrt(r2), a = sq * v_atan2(tx, ty), rr = VW * sq;            // heart
              OUT(rr * fsin(a), -rr * fcos(a)); } break;
    case 8: { float a = VW * v_atan2(tx, ty) * FM_1_PI, rr = FM_PI * fsqrt(r2);        // disc
              OUT(fsin(rr) * a, fcos(rr) * a); } break;
    case 9: { float a = v_atan2(tx, ty), rr = fsqrt(r2), r1 = fdiv(VW, rr);            // spiral
              OUT(r1 * (fcos(a) + fsin(rr)), r1 * (fsin(a) - fcos(rr))); } break;
    case 10: { float a = v_atan2(tx, ty), rr = fsqrt(r2);                              // hyperbolic
               OUT(fdiv(VW * fsin(a), rr), VW * fcos(a) * rr); } break;
    case 11: { float a = v_atan2(tx, ty), rr = fsqrt(r2);                              // diamond
               OUT(VW * fsin(a) * fcos(rr), VW * fcos(a) * fsin(rr)); } break;
    case 12: { float a = v_atan2(tx, ty), rr = fsqrt(r2);                              // ex
               float n0 = fsin(a + rr), n1 = fcos(a - rr);
               float m0 = n0 * n0 * n0 * rr, m1 = n1 * n1 * n1 * rr;
               OUT(VW * (m0 + m1), VW * (m0 - m1)); } break;
    case 13: { float a = 0.5f * v_atan2(tx, ty);                                       // julia
               if (mwc_next(r) & 1) a += FM_PI;
               float rr = VW * fsqrt(fsqrt(r2));
               OUT(rr * fcos(a), rr * fsin(a)); } break;
    case 14: { float nx = tx < 0.0f ? 2.0f : 1.0f, ny = ty < 0.0f ? 0.5f : 1.0f;      // bent
               OUT(VW * nx * tx, VW * ny * ty); } break;
    case 15: OUT(VW * (tx + xf[1] * fsin(ty * VP(0))),                                 // waves
                 VW * (ty + xf[4] * fsin(tx * VP(1)))); break;
    case 16: { float k = fdiv(2.0f * VW, fsqrt(r2) + 1.0f); OUT(k * ty, k * tx); } break;   // fisheye
    case 17: { float dx = ftan(3.0f * ty), dy = ftan(3.0f * tx);                       // popcorn
               OUT(VW * (tx + xf[2] * fsin(dx)), VW * (ty + xf[5] * fsin(dy))); } break;
    case 18: { float dx = VW * fexp(tx - 1.0f);                                        // exponential
               if (isfinite(dx)) { float dy = FM_PI * ty; OUT(dx * fcos(dy), dx * fsin(dy)); } } break;
    case 19: { float a = v_atan2(tx, ty), sa = fsin(a), rr = VW * fpow(fsqrt(r2), sa); // power
               OUT(rr * fcos(a), rr * sa); } break;
    case 20: { float a = FM_PI * tx;                                                   // cosine
               OUT(VW * fcos(a) * v_cosh(ty), -VW * fsin(a) * v_sinh(ty)); } break;
    case 21: { float dx = xf[2]; dx *= dx;                                             // rings
               float rr = fsqrt(r2), a = v_atan2(tx, ty);
               rr = VW * (v_fmod(rr + dx, 2.0f * dx) - dx + rr * (1.0f - dx));
               OUT(rr * fcos(a), rr * fsin(a)); } break;
    case 22: { float dx = xf[2]; dx *= dx * FM_PI;                                     // fan
               float dx2 = 0.5f * dx, dy = xf[5], a = v_atan2(tx, ty);
               a += (v_fmod(a + dy, dx) > dx2) ? -dx2 : dx2;
               float rr = VW * fsqrt(r2);
               OUT(rr * fcos(a), rr * fsin(a)); } break;
    case 23: { float rr = fsqrt(r2), a = v_atan2(tx, ty), bd = 0.5f * (VP(0) - VP(1)); // blob: high low waves
               rr *= VW * (VP(1) + bd * (1.0f + fsin(VP(2) * a)));
               OUT(fsin(a) * rr, fcos(a) * rr); } break;
    case 24: OUT(VW * (fsin(VP(0) * ty) - fcos(VP(1) * tx)),                           // pdj: a b c d
                 VW * (fsin(VP(2) * tx) - fcos(VP(3) * ty))); break;
    case 25: { float dy = VP(1), dx = VP(0); dx *= dx * FM_PI;                         // fan2: x y
               float dx2 = 0.5f * dx, a = v_atan2(tx, ty), rr = VW * fsqrt(r2);
               float t = a + dy - dx * truncf(fdiv(a + dy, dx));
               a += (t > dx2) ? -dx2 : dx2;
               OUT(rr * fsin(a), rr * fcos(a)); } break;
    case 26: { float dx = VP(0); dx *= dx;                                             // rings2: val
               float rr = fsqrt(r2), a = v_atan2(tx, ty);
               rr += -2.0f * dx * (float)(int)fdiv(rr + dx, 2.0f * dx) + rr * (1.0f - dx);
               OUT(VW * fsin(a) * rr, VW * fcos(a) * rr); } break;
    case 27: { float k = fdiv(2.0f * VW, fsqrt(r2) + 1.0f); OUT(k * tx, k * ty); } break;   // eyefish
    case 28: { float k = fdiv(VW, 0.25f * r2 + 1.0f); OUT(k * tx, k * ty); } break;    // bubble
    case 29: OUT(VW * fsin(tx), VW * ty); break;                                       // cylinder
    case 30: { float t = frcp(VP(2) - ty * VP(3));                                     // perspective: angle dist | mdist sin cos
               OUT(VW * VP(2) * tx * t, VW * VP(4) * ty * t); } break;
    case 31: { float a = mwc_next_01(r) * 2.0f * FM_PI, rr = VW * mwc_next_01(r);      // noise
               OUT(tx * rr * fcos(a), ty * rr * fsin(a)); } break;
    case 32: { float power = VP(1);                                                    // julian: dist power | cn
               float t_rnd = truncf(mwc_next_01(r) * fabsf(power));
               float a = fdiv(v_atan2(ty, tx) + 2.0f * FM_PI * t_rnd, power);
               float rr = VW * fpow(r2, VP(2));
               OUT(rr * fcos(a), rr * fsin(a)); } break;
    case 33: { float ang = v_atan2(ty, tx), power = VP(1);                             // juliascope
               float t_rnd = truncf(mwc_next_01(r) * fabsf(power));
               if (mwc_next(r) & 1) ang = -ang;
               float a = fdiv(2.0f * FM_PI * t_rnd + ang, power);
               float rr = VW * fpow(r2, VP(2));
               OUT(rr * fcos(a), rr * fsin(a)); } break;
    case 34: { float a = mwc_next_01(r) * 2.0f * FM_PI, rr = VW * mwc_next_01(r);      // blur
               OUT(rr * fcos(a), rr * fsin(a)); } break;
    case 35: { float a = mwc_next_01(r) * 2.0f * FM_PI, rr = v_gauss_r(VW, r);         // gaussian_blur
               OUT(rr * fcos(a), rr * fsin(a)); } break;
    case 36: { float ba = VP(0) * FM_PI * 0.5f, spin = fsin(ba), zoom = fcos(ba);      // radial_blur: angle
               float rr = v_gauss_r(VW, r), ra = fsqrt(r2);
               float a = v_atan2(ty, tx) + spin * rr, rz = zoom * rr - 1.0f;
               OUT(ra * fcos(a) + rz * tx, ra * fsin(a) + rz * ty); } break;
    case 37: { float slices = VP(1);                                                   // pie: rotation slices thickness
               float sl = truncf(mwc_next_01(r) * slices + 0.5f);
               float a = VP(0) + fdiv(2.0f * FM_PI * (sl + mwc_next_01(r) * VP(2)), slices);
               float rr = VW * mwc_next_01(r);
               OUT(rr * fcos(a), rr * fsin(a)); } break;
    case 38: { float power = VP(2) * 0.5f, b = fdiv(2.0f * FM_PI, VP(3));              // ngon: circle corners power sides
               float rf = fpow(r2, power), th = v_atan2(ty, tx);
               float phi = th - b * floorf(fdiv(th, b));
               if (phi > b * 0.5f) phi -= b;
               float amp = fdiv(VP(1) * (frcp(fcos(phi)) - 1.0f) + VP(0), rf);
               OUT(VW * tx * amp, VW * ty * amp); } break;
    case 39: { float c1 = VP(0), c2 = VP(1);                                           // curl: c1 c2
               float re = 1.0f + c1 * tx + c2 * (tx * tx - ty * ty), im = c1 * ty + 2.0f * c2 * tx * ty;
               float k = fdiv(VW, re * re + im * im);
               OUT(k * (tx * re + ty * im), k * (ty * re - tx * im)); } break;
    case 40: { float rx = VP(0), ry = VP(1);                                           // rectangles: x y
               OUT(VW * ((rx == 0.0f) ? tx : rx * (2.0f * floorf(fdiv(tx, rx)) + 1.0f) - tx),
                   VW * ((ry == 0.0f) ? ty : ry * (2.0f * floorf(fdiv(ty, ry)) + 1.0f) - ty)); } break;
    case 41: { float a = mwc_next_01(r) * VW * FM_PI, s = fsin(a);                     // arch
               OUT(VW * s, fdiv(VW * s * s, fcos(a))); } break;
    case 42: OUT(fdiv(VW * fsin(tx), fcos(ty)), VW * ftan(ty)); break;                 // tangent
    case 43: { float a = mwc_next_01(r), b = mwc_next_01(r);                           // square
               OUT(VW * (a - 0.5f), VW * (b - 0.5f)); } break;
    case 44: { float a = VW * mwc_next_01(r) * FM_PI, k = fdiv(VW, r2);                // rays
               float tr = VW * ftan(a) * k;
               OUT(tr * fcos(tx), tr * fsin(ty)); } break;
    case 45: { float rr = mwc_next_01(r) * VW * fsqrt(r2), s = fsin(rr), c = fcos(rr); // blade
               OUT(VW * tx * (c + s), VW * tx * (c - s)); } break;
    case 46: { float cr = fcos(VW * fsqrt(r2)), icr = frcp(cr);                        // secant2
               icr += (cr < 0.0f ? 1.0f : -1.0f);
               OUT(VW * tx, VW * icr); } break;
    case 48: { float s = tx * tx - ty * ty, k = VW * fsqrt(frcp(s * s));               // cross
               OUT(k * tx, k * ty); } break;
    case 49: { float twist = VP(1), rotpi = VP(0) * FM_PI;                             // disc2: rot twist
               float st = fsin(twist), ct = fcos(twist) - 1.0f;
               if (twist > 2.0f * FM_PI) { float k = 1.0f + twist - 2.0f * FM_PI; st *= k; ct *= k; }
               if (twist < -2.0f * FM_PI) { float k = 1.0f + twist + 2.0f * FM_PI; st *= k; ct *= k; }
               float t = rotpi * (tx + ty), k = fdiv(VW * v_atan2(tx, ty), FM_PI);
               OUT(k * (fsin(t) + ct), k * (fcos(t) + st)); } break;
    case 50: { float th = 0.25f * (VP(1) * v_atan2(ty, tx) + FM_PI);                   // super_shape: holes m n1 n2 n3 rnd
               float t1 = fpow(fabsf(fcos(th)), VP(3)), t2 = fpow(fabsf(fsin(th)), VP(4));
               float rnd = VP(5), d = fsqrt(r2);
               float k = fdiv(VW * ((rnd * mwc_next_01(r) + (1.0f - rnd) * d) - VP(0))
                              * fpow(t1 + t2, fdiv(-1.0f, VP(2))), d);
               OUT(k * tx, k * ty); } break;
    case 51: { float k = fdiv(VW * (mwc_next_01(r) - VP(0)) * fcos(VP(1) * v_atan2(ty, tx)), fsqrt(r2));   // flower: holes petals
               OUT(k * tx, k * ty); } break;
    case 52: { float d = fsqrt(r2), ct = fdiv(tx, d);                                  // conic: eccentricity holes
               float k = fdiv(fdiv(VW * (mwc_next_01(r) - VP(1)) * VP(0), 1.0f + VP(0) * ct), d);
               OUT(k * tx, k * ty); } break;
    case 53: { float rr = fsqrt(r2), sr = fsin(rr), cr = fcos(rr);                     // parabola: height width
               float a = mwc_next_01(r), b = mwc_next_01(r);
               OUT(VP(0) * VW * sr * sr * a, VP(1) * VW * cr * b); } break;
    case 54: { float nx = tx < 0.0f ? VP(0) : 1.0f, ny = ty < 0.0f ? VP(1) : 1.0f;     // bent2: x y
               OUT(VW * nx * tx, VW * ny * ty); } break;
    case 55: { float t = r2 + 1.0f, x2 = tx * 2.0f, ps = -FM_PI_2 * VP(0);             // bipolar: shift
               float y = 0.5f * v_atan2(2.0f * ty, r2 - 1.0f) + ps;
               if (y > FM_PI_2) y = -FM_PI_2 + v_fmod_pi(y + FM_PI_2);
               else if (y < -FM_PI_2) y = FM_PI_2 - v_fmod_pi(FM_PI_2 - y);
               OUT(VW * 0.25f * FM_2_PI * flog(fdiv(t + x2, t - x2)), VW * FM_2_PI * y); } break;
    case 56: { float rx = rintf(tx), ry = rintf(ty), fx = tx - rx, fy = ty - ry;       // boarders
               if (mwc_next_01(r) > 0.75f) {
                   OUT(VW * (fx * 0.5f + rx), VW * (fy * 0.5f + ry));
               } else if (fabsf(fx) >= fabsf(fy)) {
                   float s = fx >= 0.0f ? 0.25f : -0.25f;
                   OUT(VW * (fx * 0.5f + rx + s), VW * (fy * 0.5f + ry + s * fdiv(fy, fx)));
               } else {
                   float s = fy >= 0.0f ? 0.25f : -0.25f;
                   OUT(VW * (fx * 0.5f + rx + fdiv(fx, fy) * s), VW * (fy * 0.5f + ry + s));
               } } break;
    case 57: { float wx = VW * 1.3029400317411197908970256609023f, y2 = ty * 2.0f;     // butterfly
               float k = wx * fsqrt(fdiv(fabsf(ty * tx), tx * tx + y2 * y2));
               OUT(k * tx, k * y2); } break;
    case 58: { float cs = VP(0), ics = frcp(cs);                                       // cell: size
               float cx = floorf(tx * ics), cy = floorf(ty * ics);
               float dx = tx - cx * cs, dy = ty - cy * cs;
               cx = cx >= 0.0f ? cx * 2.0f : -(2.0f * cx + 1.0f);
               cy = cy >= 0.0f ? cy * 2.0f : -(2.0f * cy + 1.0f);
               OUT(VW * (dx + cx * cs), -VW * (dy + cy * cs)); } break;
    case 59: { float a = v_atan2(ty, tx), lnr = 0.5f * flog(r2), power = frcp(VP(1));  // cpow: i power r
               float va = 2.0f * FM_PI * power, vc = VP(2) * power, vd = VP(0) * power;
               float ang = vc * a + vd * lnr + va * floorf(power * mwc_next_01(r));
               float m = VW * fexp(vc * lnr - vd * a);
               OUT(m * fcos(ang), m * fsin(ang)); } break;
    case 60: OUT(VW * (tx + VP(0) * fexp(-ty * ty * VP(4))),                           // curve: xamp xlength yamp ylength | x2 y2
                 VW * (ty + VP(2) * fexp(-tx * tx * VP(5)))); break;
    case 61: { float tmp = r2 + 1.0f, tmp2 = 2.0f * tx;                                // edisc
               float xmax = (fsqrt(tmp + tmp2) + fsqrt(tmp - tmp2)) * 0.5f;
               float a1 = flog(xmax + fsqrt(xmax - 1.0f)), a2 = -acosf(fdiv(tx, xmax)), nw = fdiv(VW, 11.57034632f);
               float snv = fsin(a1), csv = fcos(a1);
               if (ty > 0.0f) snv = -snv;
               OUT(nw * v_cosh(a2) * csv, nw * v_sinh(a2) * snv); } break;
    case 62: { float tmp = r2 + 1.0f, x2 = 2.0f * tx;                                  // elliptic
               float xmax = 0.5f * (fsqrt(tmp + x2) + fsqrt(tmp - x2));
               float a = fdiv(tx, xmax), b = 1.0f - a * a, ssx = xmax - 1.0f, nw = fdiv(VW, FM_PI_2);
               b = b < 0.0f ? 0.0f : fsqrt(b);
               ssx = ssx < 0.0f ? 0.0f : fsqrt(ssx);
               float l = nw * flog(xmax + ssx);
               OUT(nw * v_atan2(a, b), ty > 0.0f ? l : -l); } break;
    case 63: { float a = v_atan2(ty, tx), lnr = 0.5f * flog(r2);                       // escher: beta
               float vc = 0.5f * (1.0f + fcos(VP(0))), vd = 0.5f * fsin(VP(0));
               float m = VW * fexp(vc * lnr - vd * a), n = vc * a + vd * lnr;
               OUT(m * fcos(n), m * fsin(n)); } break;
    case 64: { float ex = fexp(tx) * 0.5f, enx = fdiv(0.25f, ex), sn = fsin(ty), cn = fcos(ty);   // foci
               float t = fdiv(VW, ex + enx - cn);
               OUT(t * (ex - enx), t * sn); } break;
    case 65: { float lx = VP(3), ly = VP(4), x = tx - lx, y = ty + ly, rr = fsqrt(x * x + y * y);  // lazysusan: space spin twist x y
               if (rr < VW) {
                   float a = v_atan2(y, x) + VP(1) + VP(2) * (VW - rr);
                   OUT(VW * (rr * fcos(a) + lx), VW * (rr * fsin(a) - ly));
               } else {
                   rr = 1.0f + fdiv(VP(0), rr);
                   OUT(VW * (rr * x + lx), VW * (rr * y - ly));
               } } break;
    case 66: { float w2 = VW * VW;                                                     // loonie
               float k = (r2 < w2) ? VW * fsqrt(fdiv(w2, r2) - 1.0f) : VW;
               OUT(k * tx, k * ty); } break;
    case 67: { float g = mwc_next_01(r); g += mwc_next_01(r); g += mwc_next_01(r); g += mwc_next_01(r);   // pre_blur
               g = VW * (g - 2.0f);
               float a = mwc_next_01(r) * 2.0f * FM_PI;
               tx += g * fcos(a); ty += g * fsin(a); } break;
    case 68: { float mx = VP(0), my = VP(1), xr = 2.0f * mx, yr = 2.0f * my;           // modulus: x y
               float ax = (tx > mx) ? VW * (-mx + v_fmod(tx + mx, xr)) : (tx < -mx) ? VW * (mx - v_fmod(mx - tx, xr)) : VW * tx;
               float ay = (ty > my) ? VW * (-my + v_fmod(ty + my, yr)) : (ty < -my) ? VW * (my - v_fmod(my - ty, yr)) : VW * ty;
               OUT(ax, ay); } break;
    case 69: { float tpf = 2.0f * FM_PI * VP(2);                                       // oscope: amplitude damping frequency separation
               float t = VP(0) * fexp(-fabsf(tx) * VP(1)) * fcos(tpf * tx) + VP(3);
               OUT(VW * tx, (fabsf(ty) <= t) ? -VW * ty : VW * ty); } break;
    case 70: { float p2v = fdiv(VW, FM_PI);                                            // polar2
               OUT(p2v * v_atan2(tx, ty), 0.5f * p2v * flog(r2)); } break;
    case 71: OUT(VW * (tx + VP(1) * fsin(ftan(ty * VP(0)))),                           // popcorn2: c x y
                 VW * (ty + VP(2) * fsin(ftan(tx * VP(0))))); break;
    case 72: { float k = frcp(fsqrt(r2) * (r2 + frcp(VW))); OUT(tx * k, ty * k); } break;   // scry
    case 73: { float sx2 = VP(0) * VP(0), sy2 = VP(2) * VP(2);                         // separation: x xinside y yinside
               float ax = tx > 0.0f ? VW * (fsqrt(tx * tx + sx2) - tx * VP(1)) : -VW * (fsqrt(tx * tx + sx2) + tx * VP(1));
               float ay = ty > 0.0f ? VW * (fsqrt(ty * ty + sy2) - ty * VP(3)) : -VW * (fsqrt(ty * ty + sy2) + ty * VP(3));
               OUT(ax, ay); } break;
    case 74: OUT((fcos(ty * VP(1) * FM_PI) >= 0.0f) ? VW * tx : -VW * tx,              // split: xsize ysize
                 (fcos(tx * VP(0) * FM_PI) >= 0.0f) ? VW * ty : -VW * ty); break;
    case 75: OUT(VW * (tx + copysignf(VP(0), tx)), VW * (ty + copysignf(VP(1), ty))); break;   // splits: x y
    case 76: { float rx = floorf(tx + 0.5f), fx = tx - rx;                             // stripes: space warp
               OUT(VW * (fx * (1.0f - VP(0)) + rx), VW * (ty + fx * fx * VP(1))); } break;
    case 77: { float rr = fsqrt(r2), a = v_atan2(ty, tx) + VP(3) * rr, wc = VP(1), wa = VP(0);   // wedge: angle count hole swirl
               float c = floorf((wc * a + FM_PI) * FM_1_PI * 0.5f);
               a = a * (1.0f - wa * wc * FM_1_PI * 0.5f) + c * wa;
               rr = VW * (rr + VP(2));
               OUT(rr * fcos(a), rr * fsin(a)); } break;
    case 80: { float rr = fsqrt(r2), a = v_atan2(ty, tx);                              // whorl: inside outside
               a += fdiv(rr < VW ? VP(0) : VP(1), VW - rr);
               OUT(VW * rr * fcos(a), VW * rr * fsin(a)); } break;
    case 81: OUT(VW * (tx + VP(2) * fsin(ty * VP(0))), VW * (ty + VP(3) * fsin(tx * VP(1)))); break;   // waves2: freqx freqy scalex scaley
    case 82: { float e = fexp(tx); OUT(VW * e * fcos(ty), VW * e * fsin(ty)); } break;  // exp
    case 83: OUT(VW * 0.5f * flog(r2), VW * v_atan2(ty, tx)); break;                   // log
    case 84: OUT(VW * fsin(tx) * v_cosh(ty), VW * fcos(tx) * v_sinh(ty)); break;         // sin
    case 85: OUT(VW * fcos(tx) * v_cosh(ty), -VW * fsin(tx) * v_sinh(ty)); break;        // cos
    case 86: { float d = frcp(fcos(2.0f * tx) + v_cosh(2.0f * ty));                     // tan
               OUT(VW * d * fsin(2.0f * tx), VW * d * v_sinh(2.0f * ty)); } break;
    case 87: { float d = fdiv(2.0f, fcos(2.0f * tx) + v_cosh(2.0f * ty));               // sec
               OUT(VW * d * fcos(tx) * v_cosh(ty), VW * d * fsin(tx) * v_sinh(ty)); } break;
    case 88: { float d = fdiv(2.0f, v_cosh(2.0f * ty) - fcos(2.0f * tx));               // csc
               OUT(VW * d * fsin(tx) * v_cosh(ty), -VW * d * fcos(tx) * v_sinh(ty)); } break;
    case 89: { float d = frcp(v_cosh(2.0f * ty) - fcos(2.0f * tx));                     // cot
               OUT(VW * d * fsin(2.0f * tx), -VW * d * v_sinh(2.0f * ty)); } break;
    case 90: OUT(VW * v_sinh(tx) * fcos(ty), VW * v_cosh(tx) * fsin(ty)); break;         // sinh
    case 91: OUT(VW * v_cosh(tx) * fcos(ty), VW * v_sinh(tx) * fsin(ty)); break;         // cosh
    case 92: { float d = frcp(fcos(2.0f * ty) + v_cosh(2.0f * tx));                     // tanh
               OUT(VW * d * v_sinh(2.0f * tx), VW * d * fsin(2.0f * ty)); } break;
    case 93: { float d = fdiv(2.0f, fcos(2.0f * ty) + v_cosh(2.0f * tx));               // sech
               OUT(VW * d * fcos(ty) * v_cosh(tx), -VW * d * fsin(ty) * v_sinh(tx)); } break;
    case 94: { float d = fdiv(2.0f, v_cosh(2.0f * tx) - fcos(2.0f * ty));               // csch
               OUT(VW * d * v_sinh(tx) * fcos(ty), -VW * d * v_cosh(tx) * fsin(ty)); } break;
    case 95: { float d = frcp(v_cosh(2.0f * tx) - fcos(2.0f * ty));                     // coth
               OUT(VW * d * v_sinh(2.0f * tx), VW * d * fsin(2.0f * ty)); } break;
    case 97: { float xpw = tx + VW, xmw = tx - VW, y2 = ty * ty;                       // flux: spread
               float ar = VW * (2.0f + VP(0)) * fsqrt(fdiv(fsqrt(y2 + xpw * xpw), fsqrt(y2 + xmw * xmw)));
               float aa = (v_atan2(ty, xmw) - v_atan2(ty, xpw)) * 0.5f;
               OUT(ar * fcos(aa), ar * fsin(aa)); } break;
    case 98: { float ima = VP(0), imb = VP(1), imc = VP(2), imd = VP(3);               // mobius: im_a..im_d re_a..re_d
               float rea = VP(4), reb = VP(5), rec = VP(6), red = VP(7);
               float re_u = rea * tx - ima * ty + reb, im_u = rea * ty + ima * tx + imb;
               float re_v = rec * tx - imc * ty + red, im_v = rec * ty + imc * tx + imd;
               float k = fdiv(VW, re_v * re_v + im_v * im_v);
               OUT(k * (re_u * re_v + im_u * im_v), k * (im_u * re_v - re_u * im_v)); } break;
    default: return false;
    }
    return true;
}

__device__ __noinline__ VarIO apply_variation_slow(int id, float w, const float *__restrict__ v,
                                                   const float *__restrict__ xf, VarIO io, uint32_t mul)
{
    mwc_t r = {mul, io.state, io.carry};
    apply_variation_body(id, w, v, xf, io.tx, io.ty, io.ox, io.oy, r);
    io.state = r.state; io.carry = r.carry;
    return io;
}

// id / w: the variation's number and weight; v points at its parameters (include/flame_hip.h (5)).
__device__ __forceinline__ void apply_variation(int id, float w, const float *__restrict__ v,
                                                const float *__restrict__ xf,
                                                float &tx, float &ty, float &ox, float &oy, mwc_t &r)
{
    // Variations with an inline fast path; everything else goes through one out-of-line call.
    // The fast set is tested first (one scalar bit test), so the switch below has no default
    // edge: with one, the scalar control flow gets structurised into a chain of flag registers
    // that costs more than the simple variations themselves.
    constexpr uint32_t kFast = (1u << 0) | (1u << 1) | (1u << 2) | (1u << 3) | (1u << 4) | (1u << 14) | (1u << 27) | (1u << 28);
    if ((uint32_t)id >= 32u || !((kFast >> id) & 1u)) {
        VarIO io = {tx, ty, ox, oy, r.state, r.carry};
        io = apply_variation_slow(id, w, v, xf, io, r.mul);
        tx = io.tx; ty = io.ty; ox = io.ox; oy = io.oy; r.state = io.state; r.carry = io.carry;
        return;
    }
    const float r2 = fmaf(tx, tx, ty * ty);
    switch (id) {
    case 0: { OUT(tx * VW, ty * VW); } break;                                              // linear
    case 2: { float k = fdiv(VW, r2); OUT(tx * k, ty * k); } break;                        // spherical
    case 3: { float s = fsin(r2), c = fcos(r2);                                            // swirl
              OUT(VW * (s * tx - c * ty), VW * (c * tx + s * ty)); } break;
    case 1: { OUT(VW * fsin(tx), VW * fsin(ty)); } break;                                  // sinusoidal
    case 4: { float k = fdiv(VW, fsqrt(r2));                                               // horseshoe
              OUT(k * (tx - ty) * (tx + ty), 2.0f * tx * ty * k); } break;
    case 27: { float k = fdiv(2.0f * VW, fsqrt(r2) + 1.0f); OUT(k * tx, k * ty); } break;  // eyefish
    case 28: { float k = fdiv(VW, 0.25f * r2 + 1.0f); OUT(k * tx, k * ty); } break;        // bubble
    case 14: { float nx = tx < 0.0f ? 2.0f : 1.0f, ny = ty < 0.0f ? 0.5f : 1.0f;           // bent
               OUT(VW * nx * tx, VW * ny * ty); } break;
    default: __builtin_unreachable();
    }
}
#undef VW
#undef VP
#undef OUT
