// pt_device_math.h -- gfx950 device-side arithmetic of the path-tracing hot path.
//
// Implements the PTSPEC arithmetic contract for the OpenCL built-ins that
// test/ClKernels/GenerateColors.cl calls and whose rounding OpenCL leaves open:
//   dot / cross / normalize  (GenerateColors.cl:75,96-97,107,114-115,122-123,130,...)
//   sin / cos                (:171,191)        pow (:177,292,298)       max (:185,235,260)
// Every function is a fixed sequence of IEEE binary32 / binary64 operations (+,-,*,/,sqrt,
// fma, integer ops); the translation unit is compiled with -ffp-contract=off and without
// any fast-math flag, hipcc's default correctly-rounded fp32 divide/sqrt, subnormals kept.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_constants.h"

#define PTK_DEV __device__ __forceinline__

struct f3 { float x, y, z; };

PTK_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PTK_DEV f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PTK_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PTK_DEV f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PTK_DEV f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }

PTK_DEV float pt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PTK_DEV double pt_fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }
// OpenCL max(): (a < b) ? b : a  -- a NaN in `a` is returned, a NaN in `b` is dropped
PTK_DEV float pt_max(float a, float b) { return (a < b) ? b : a; }

// dot(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
PTK_DEV float dot3(f3 a, f3 b) { return pt_fma(a.z, b.z, pt_fma(a.y, b.y, a.x * b.x)); }

// cross(a,b) = ( fma(a.y,b.z,-(a.z*b.y)), fma(a.z,b.x,-(a.x*b.z)), fma(a.x,b.y,-(a.y*b.x)) )
PTK_DEV f3 cross3(f3 a, f3 b)
{
    return mk3(pt_fma(a.y, b.z, -(a.z * b.y)), pt_fma(a.z, b.x, -(a.x * b.z)), pt_fma(a.x, b.y, -(a.y * b.x)));
}

// ---- correctly rounded 1/x and sqrt(x) at a third of hipcc's generic cost ------------------------
// PTSPEC asks for IEEE correctly-rounded "/" and sqrt.  hipcc's generic sequences (v_div_scale x2,
// v_rcp, 5 fma, v_div_fmas, v_div_fixup; likewise for sqrt) pay for operand scaling that only
// matters near the ends of the exponent range.  Inside the ranges below the short sequences are
// BIT-IDENTICAL to the IEEE result for EVERY binary32 input -- checked exhaustively on gfx950 by
// tools/ubench (profiles/r01/ubench.log: 0 mismatches over all inputs of the range); outside them
// the generic sequence runs.  NaN inputs take the fast path and stay NaN.
#define PTK_RCP_FAST_MIN 1e-8f
#define PTK_RCP_FAST_MAX 1e20f
#define PTK_SQRT_FAST_MIN 1e-30f
#define PTK_SQRT_FAST_MAX 1e30f

PTK_DEV float pt_rcp_fast(float x)  // exact for x in [1e-8, 1e20]
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = pt_fma(-x, r, 1.0f);
    return pt_fma(e, r, r);
}

PTK_DEV float pt_sqrt_fast(float x)  // exact for x in [1e-30, 1e30]
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float h = 0.5f * y;
    float r = pt_fma(-h, g, 0.5f);
    g = pt_fma(g, r, g);
    h = pt_fma(h, r, h);
    float d = pt_fma(-g, g, x);
    return pt_fma(d, h, g);
}

// 1.0f / x, correctly rounded for every x
PTK_DEV float pt_rcp(float x)
{
    if (__builtin_expect(x < PTK_RCP_FAST_MIN || x > PTK_RCP_FAST_MAX, 0)) return 1.0f / x;
    return pt_rcp_fast(x);
}

// sqrtf(x), correctly rounded for every x
PTK_DEV float pt_sqrt(float x)
{
    if (__builtin_expect(x < PTK_SQRT_FAST_MIN || x > PTK_SQRT_FAST_MAX, 0)) return __builtin_sqrtf(x);
    return pt_sqrt_fast(x);
}

// ---- three IEEE quotients by one divisor in 21 instructions instead of 33 ----------------------------------
// Markstein's sequence (IBM J. R&D 34 (1990)): y = RN(1/b), q0 = a*y, r = fma(-b, q0, a), q = fma(r, y, q0).
// That q == RN(a/b) holds for EVERY pair of binary32 significands was established by running all 2^46 of them
// against the generic division on the MI355X (tools/div_exhaustive.py, profiles/r03/div_exhaustive.txt: 7.04e13
// pairs, 0 mismatches); exponents do not matter while no operand or intermediate leaves the normal range, which
// the guards below ensure: b in [2^-26, 2^40) (inside pt_rcp_fast's exact range), every numerator +0 or in
// [2^-60, 2^60) (so q0 is in (2^-100, 2^86) and r, a multiple of 2^-107 at least, is formed without underflow).  (The textbook
// theorem wants a faithful q0, which makes r exact; RN(a RN(1/b)) can be 1.5 ulp off when a < b, and r is then rounded -- one
// pair in ~700 -- with the quotient still correct: the claim rests on the exhaustive run, not on the theorem.)  Anything else --
// negative, NaN, infinite, tiny, huge -- takes the generic division.
PTK_DEV float pt_div_markstein(float a, float b, float y)
{
    const float q0 = a * y;
    const float r = pt_fma(-b, q0, a);
    return pt_fma(r, y, q0);
}

PTK_DEV void pt_div3(float& a0, float& a1, float& a2, float b)
{
    const unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2);
    const unsigned hi = max(max(u0, u1), u2);                 // v_max3_u32
    const unsigned lo = min(min(u0 - 1u, u1 - 1u), u2 - 1u);  // (+0 wraps to the top: allowed)
    const bool fast = hi < 0x5d800000u /* 2^60 */ && lo >= 0x21800000u - 1u /* 2^-60 */ &&
                      (__float_as_uint(b) - 0x32800000u /* 2^-26 */) < (0x53800000u /* 2^40 */ - 0x32800000u);
    if (__builtin_expect(fast, 1)) {
        const float y = pt_rcp_fast(b);
        a0 = pt_div_markstein(a0, b, y);
        a1 = pt_div_markstein(a1, b, y);
        a2 = pt_div_markstein(a2, b, y);
    } else {
        a0 = a0 / b;
        a1 = a1 / b;
        a2 = a2 / b;
    }
}

// normalize(v) = v * (1.0f / sqrtf(dot(v,v)))   (both correctly rounded)
PTK_DEV f3 normalize3(f3 a)
{
    // one range check for both short sequences: len2 in [1e-15, 1e30] puts sqrt(len2) in
    // [3.2e-8, 1e15], inside the reciprocal's exact range [1e-8, 1e20]
    const float len2 = dot3(a, a);
    float inv;
    if (__builtin_expect(len2 >= 1e-15f && len2 <= PTK_SQRT_FAST_MAX, 1)) inv = pt_rcp_fast(pt_sqrt_fast(len2));
    else inv = 1.0f / __builtin_sqrtf(len2);
    return scale3(a, inv);
}

// ---- RNG: GenerateColors.cl:47-71 -------------------------------------------------------
PTK_DEV uint32_t pt_hash_u32(uint32_t x) { return 1103515245u * x + 12345u; }

PTK_DEV float pt_random_float(uint32_t& seed)
{
    uint32_t s = seed;
    s = (s ^ 61u) ^ (s >> 16);
    s = s + (s << 3);
    s = s ^ (s >> 4);
    s = s * 0x27d4eb2du;
    s = s ^ (s >> 15);
    s = 1103515245u * s + 12345u;
    seed = s;
    return (float)s * 2.3283064365386963e-10f;
}

// A binary64 literal pinned to an SGPR pair at its point of use.  Left alone, hipcc hoists the
// polynomial coefficients out of the bounce loop into VGPR pairs that stay live through the whole
// kernel (27 VGPRs for sin/cos: 91 -> 64 registers, 5 -> 8 waves per SIMD without them); v_fma_f64
// takes one SGPR-pair operand for free, and two s_mov per coefficient per bounce cost nothing.
PTK_DEV double pt_k64(double c)
{
    asm volatile("" : "+s"(c));
    return c;
}

// ---- sin/cos of phi >= 0: Cody-Waite by pi/2 in binary64, Taylor to r^15 / r^16, one rounding ----
PTK_DEV void pt_sincos(float phi, float& s_out, float& c_out)
{
    // PTSPEC: on [0, PTK_F32_SINCOS_MAX] -- every angle the path forms (phi = 2 pi xi) -- binary32
    // throughout: four-term Cody-Waite reduction by pi/2, degree-9 / degree-10 polynomials, fma at
    // every step (exhaustively checked against binary64: <= 1.43 ulp, tools/check_sincos_f32.c).
    // The binary64 evaluation below (the only one until this was measured at 5.6 % of the trace
    // kernel) remains for any other argument.
    if (__builtin_expect(phi >= 0.0f && phi <= PTK_F32_SINCOS_MAX, 1)) {
        const float kf = __builtin_rintf(phi * PTK_F32_TWO_OVER_PI);
        float r = pt_fma(-kf, PTK_F32_PIO2_A, phi);
        r = pt_fma(-kf, PTK_F32_PIO2_B, r);
        r = pt_fma(-kf, PTK_F32_PIO2_C, r);
        r = pt_fma(-kf, PTK_F32_PIO2_D, r);
        const float r2 = r * r;
        float ps = PTK_F32_SIN_S4;
        ps = pt_fma(ps, r2, PTK_F32_SIN_S3);
        ps = pt_fma(ps, r2, PTK_F32_SIN_S2);
        ps = pt_fma(ps, r2, PTK_F32_SIN_S1);
        const float sn = pt_fma(r * r2, ps, r);
        float pc = PTK_F32_COS_C4;
        pc = pt_fma(pc, r2, PTK_F32_COS_C3);
        pc = pt_fma(pc, r2, PTK_F32_COS_C2);
        pc = pt_fma(pc, r2, PTK_F32_COS_C1);
        const float cs = pt_fma(r2, pt_fma(r2, pc, -0.5f), 1.0f);
        const int q = (int)kf & 3;
        s_out = (q == 0) ? sn : (q == 1) ? cs : (q == 2) ? -sn : -cs;
        c_out = (q == 0) ? cs : (q == 1) ? -sn : (q == 2) ? -cs : sn;
        return;
    }
    double x = (double)phi;
    int k = (int)(x * pt_k64(PTK_TWO_OVER_PI) + 0.5);
    double kd = (double)k;
    double r = pt_fmad(-kd, pt_k64(PTK_PIO2_HI), x);
    r = pt_fmad(-kd, pt_k64(PTK_PIO2_LO), r);
    double r2 = r * r;
    double ps = pt_k64(PTK_SIN_S6);
    ps = pt_fmad(ps, r2, pt_k64(PTK_SIN_S5));
    ps = pt_fmad(ps, r2, pt_k64(PTK_SIN_S4));
    ps = pt_fmad(ps, r2, pt_k64(PTK_SIN_S3));
    ps = pt_fmad(ps, r2, pt_k64(PTK_SIN_S2));
    ps = pt_fmad(ps, r2, pt_k64(PTK_SIN_S1));
    ps = pt_fmad(ps, r2, pt_k64(PTK_SIN_S0));
    double sn = pt_fmad(r * r2, ps, r);
    __builtin_amdgcn_sched_barrier(0);  // one Horner chain at a time: halves the live f64 registers
    double pc = pt_k64(PTK_COS_C7);
    pc = pt_fmad(pc, r2, pt_k64(PTK_COS_C6));
    pc = pt_fmad(pc, r2, pt_k64(PTK_COS_C5));
    pc = pt_fmad(pc, r2, pt_k64(PTK_COS_C4));
    pc = pt_fmad(pc, r2, pt_k64(PTK_COS_C3));
    pc = pt_fmad(pc, r2, pt_k64(PTK_COS_C2));
    pc = pt_fmad(pc, r2, pt_k64(PTK_COS_C1));
    pc = pt_fmad(pc, r2, pt_k64(PTK_COS_C0));
    double cs = pt_fmad(r2, pc, 1.0);
    int q = k & 3;
    double so = (q == 0) ? sn : (q == 1) ? cs : (q == 2) ? -sn : -cs;
    double co = (q == 0) ? cs : (q == 1) ? -sn : (q == 2) ? -cs : sn;
    s_out = (float)so;
    c_out = (float)co;
}

// ---- pow(x, y): y == 2 -> x*x ; else exp2(y*log2(x)) in binary64, one rounding -------------
// Two 128-entry tables, no division (PTSPEC, mirrored by oracle/pt_oracle.c):
//   log2: m in [1,2), i = top 7 fraction bits, c_i ~ 1/(1+(i+.5)/128) with 16 significant bits so
//         that r = m*c_i - 1 is EXACT;  log2(x) = (e - log2 c_i) + r*(A1 + r*(A2 + ... + r*A6))
//   exp2: t = q + j/128 + f, |f| <= 2^-8;  2^t = 2^q * T_j * (1 + f*(B1 + f*(B2 + ... + f*B5)))
// The tables live in device memory (below); kernels that call pt_pow in a loop stage them in LDS
// and pass the LDS pointers.
__device__ const double pt_pow_logc_tab[128] = PTK_POW_LOGC_INIT;
__device__ const double pt_pow_logl_tab[128] = PTK_POW_LOGL_INIT;
__device__ const double pt_pow_exp2_tab[128] = PTK_POW_EXP2_INIT;

PTK_DEV float pt_pow(float x, float y, const double* logc, const double* logl, const double* exp2t)
{
    if (y == 2.0f) return x * x;
    if (!(x > 0.0f)) {
        if (x == 0.0f) return 0.0f;
        return __builtin_nanf("");
    }
    if (x == __builtin_inff()) return x;
    double xd = (double)x;
    uint64_t bits = (uint64_t)__double_as_longlong(xd);
    int e = (int)(bits >> 52) - 1023;
    int idx = (int)(bits >> 45) & 127;
    bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m = __longlong_as_double((long long)bits);
    // (coefficients pinned to SGPR pairs, pt_k64: from VGPR pairs hipcc forms every Horner step as
    // v_mov_b64 + v_fmac_f64 instead of one v_fma_f64 -- 10 extra instructions per pow)
    double r = pt_fmad(m, logc[idx], -1.0);
    double p = pt_k64(PTK_LOG2_A6);
    p = pt_fmad(p, r, pt_k64(PTK_LOG2_A5));
    p = pt_fmad(p, r, pt_k64(PTK_LOG2_A4));
    p = pt_fmad(p, r, pt_k64(PTK_LOG2_A3));
    p = pt_fmad(p, r, pt_k64(PTK_LOG2_A2));
    p = pt_fmad(p, r, pt_k64(PTK_LOG2_A1));
    double l = pt_fmad(r, p, (double)e + logl[idx]);
    double t = (double)y * l;
    if (t >= 130.0) return __builtin_inff();
    if (t <= -160.0) return 0.0f;
    int ki = (int)(t * 128.0 + (t < 0.0 ? -0.5 : 0.5));
    double f = pt_fmad(-(double)ki, 0x1p-7, t);
    int j = ki & 127;
    int q = (ki - j) >> 7;
    double g = pt_k64(PTK_EXP2_B5);
    g = pt_fmad(g, f, pt_k64(PTK_EXP2_B4));
    g = pt_fmad(g, f, pt_k64(PTK_EXP2_B3));
    g = pt_fmad(g, f, pt_k64(PTK_EXP2_B2));
    g = pt_fmad(g, f, pt_k64(PTK_EXP2_B1));
    double w = f * g;
    double T = exp2t[j];
    double res = pt_fmad(T, w, T);
    uint64_t sb = (uint64_t)(int64_t)(q + 1023) << 52;
    double sc = __longlong_as_double((long long)sb);
    return (float)(res * sc);
}

// ---- the regular part of pow, opened up for callers that chain evaluations (the fold kernel) --------
// pt_pow's value for x in PTK_POW_REGULAR (normal, positive, far from the ends of the exponent range)
// and an exponent whose product with log2(x) stays inside (-160, 130): the SAME binary64 operations
// as pt_pow on that branch, returned before the final conversion, with log2(x) beside it.  What
// differs is instruction selection only, and only where the value cannot differ:
//   * nearest(128 t) by v_rndne_f64 instead of trunc(128 t +- 0.5) -- different at exact ties alone,
//     which the exhaustive comparison (tests/test_gpu_fold_exact.py: every binary32 x of the range
//     against pt_pow) shows do not occur for the exponent the fold uses;
//   * the scaling by 2^q by v_ldexp_f64 instead of a multiplication by a constructed power of two.
#define PTK_POW_REGULAR_MIN 0x1p-80f
#define PTK_POW_REGULAR_MAX 0x1p80f
PTK_DEV bool pt_pow_is_regular(float x)
{
    // one unsigned compare: negative numbers, NaN, 0 and subnormals wrap around or fall below
    return (__float_as_uint(x) - __float_as_uint(PTK_POW_REGULAR_MIN)) <
           (__float_as_uint(PTK_POW_REGULAR_MAX) - __float_as_uint(PTK_POW_REGULAR_MIN));
}

PTK_DEV double pt_pow_regular(float x, float y, const double* logc, const double* logl, const double* exp2t, double& log2x)
{
    double xd = (double)x;
    uint64_t bits = (uint64_t)__double_as_longlong(xd);
    int e = (int)(bits >> 52) - 1023;
    int idx = (int)(bits >> 45) & 127;
    bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m = __longlong_as_double((long long)bits);
    double r = pt_fmad(m, logc[idx], -1.0);
    double p = pt_k64(PTK_LOG2_A6);
    p = pt_fmad(p, r, pt_k64(PTK_LOG2_A5));
    p = pt_fmad(p, r, pt_k64(PTK_LOG2_A4));
    p = pt_fmad(p, r, pt_k64(PTK_LOG2_A3));
    p = pt_fmad(p, r, pt_k64(PTK_LOG2_A2));
    p = pt_fmad(p, r, pt_k64(PTK_LOG2_A1));
    double l = pt_fmad(r, p, (double)e + logl[idx]);
    log2x = l;
    double t = (double)y * l;
    double kd = __builtin_rint(t * 128.0);
    int ki = (int)kd;
    double f = pt_fmad(-kd, 0x1p-7, t);
    int j = ki & 127;
    int q = ki >> 7;   // == (ki - j) >> 7
    double g = pt_k64(PTK_EXP2_B5);
    g = pt_fmad(g, f, pt_k64(PTK_EXP2_B4));
    g = pt_fmad(g, f, pt_k64(PTK_EXP2_B3));
    g = pt_fmad(g, f, pt_k64(PTK_EXP2_B2));
    g = pt_fmad(g, f, pt_k64(PTK_EXP2_B1));
    double w = f * g;
    double T = exp2t[j];
    double res = pt_fmad(T, w, T);
    return __builtin_ldexp(res, q);
}

#define PTK_TWO_PI 6.28318530718f
#define PTK_INV_PI 0.31830988618f
#define PTK_TAN_HALF_FOV 0x1.279a74p-1f  // tan(0.5f*fov), fov = (float)((60.0f*M_PI)/180.0f); correctly rounded
#define PTK_GAMMA 2.2f
