// cap_unscaled.h — the IEEE division and square root of the arithmetic contract without the scaling steps of hipcc's expansions.
// Device code only (gfx950).
//
// hipcc expands a / b into v_div_scale x 2, v_rcp, seven FMAs, v_div_fmas, v_div_fixup (~47 SIMD cycles at the measured class costs,
// docs/experiments.md (60)) and sqrtf(x) into a conditional 2^32 pre-scale, v_sqrt, two one-ulp candidates with their residuals, two
// selects, the un-scale and a class fix-up (~63).  The scale instructions only act on extreme exponents -- a denormal or huge
// denominator, a numerator below 2^-103, a quotient that would be denormal or whose exponents differ by 96 or more (CDNA ISA,
// V_DIV_SCALE_F32); x < 2^-96 for the square root -- and the fix-ups only on zeros, infinities and NaNs.  Everywhere else they
// pass their operands through, and what remains is the sequence below: the same instructions on the same operands, so the same bits
// (29 / 36 cycles).  These forms are therefore ONLY called on operands known to be in range (the *_ok predicates; the callers take
// the plain `/` and sqrtf() for a tile or wave that fails -- the same result by definition).  cap_debug_get(CAP_DEBUG_SELFTEST_DIV)
// compares both forms on the device, bit for bit, over every float of the ranges.
#pragma once

#include "cap_math.h"

namespace cap
{
__device__ __forceinline__ float div_unscaled(float a, float b)
{
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e0 = fmaf(-b, r0, 1.0f);
    const float r  = fmaf(e0, r0, r0);
    const float q0 = a * r;
    const float e1 = fmaf(-b, q0, a);
    const float q1 = fmaf(e1, r, q0);
    const float e2 = fmaf(-b, q1, a);
    return fmaf(e2, r, q1);
}
__device__ __forceinline__ float sqrt_unscaled(float x)
{
    float       s  = __builtin_amdgcn_sqrtf(x);
    const float sd = u2f(f2u(s) - 1u), su = u2f(f2u(s) + 1u);  // the neighbours one ulp below and above
    const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
    s = (0.0f >= rd) ? sd : s;
    s = (0.0f < ru) ? su : s;
    return s;
}
// Operand ranges the unscaled forms are used on (NaN fails every test).
//   denominators and square-root arguments that must be positive: [2^-40, 2^40]
//   numerators: 0 or a magnitude in [2^-80, 2^41]
//   square-root arguments: 0 or [2^-60, 2^60]
__device__ __forceinline__ bool den_ok(float x) { return x >= 9.094947e-13f && x <= 1.0995116e12f; }
__device__ __forceinline__ bool num_ok(float x) { return x == 0.0f || (fabsf(x) >= 8.2718061e-25f && fabsf(x) <= 2.1990233e12f); }
__device__ __forceinline__ bool sqrt_ok(float x) { return x == 0.0f || (x >= 8.6736174e-19f && x <= 1.1529215e18f); }

template <bool NS>
__device__ __forceinline__ float div_t(float a, float b)
{
    return NS ? div_unscaled(a, b) : a / b;
}
template <bool NS>
__device__ __forceinline__ float sqrt_t(float x)
{
    return NS ? sqrt_unscaled(x) : sqrtf(x);
}
// cap_math.h normalize3: v * (1 / sqrt(v.v)); `ok` collects whether the unscaled forms may be used on this lane's operands
template <bool NS>
__device__ __forceinline__ v3 normalize3_t(v3 v, bool& ok)
{
    const float d = dot3(v, v);
    if (NS) ok = ok && den_ok(d);  // (sqrt of [2^-40, 2^40] is in [2^-20, 2^20]: the division is in range as well)
    const float inv = div_t<NS>(1.0f, sqrt_t<NS>(d));
    return v * inv;
}
}  // namespace cap
