// cap_unscaled.h — the IEEE division of the arithmetic contract without the scaling steps of hipcc's expansion.  Device code only (gfx950).
//
// hipcc expands a / b into v_div_scale x 2, v_rcp, seven FMAs, v_div_fmas, v_div_fixup (~47 SIMD cycles at the measured class costs,
// docs/experiments.md (60)).  The scale instructions only act on extreme exponents -- a denormal or huge denominator, a numerator
// below 2^-103, a quotient that would be denormal or whose exponents differ by 96 or more (CDNA ISA, V_DIV_SCALE_F32) -- and the
// fix-up only on zeros, infinities and NaNs.  Everywhere else they pass their operands through, and what remains is the sequence
// below: the same instructions on the same operands, so the same bits (29 cycles).  It is therefore ONLY called on operands known to
// be in range -- |a| = 0 or in [2^-80, 2^41], b in [2^-40, 2^41) (a sigma <= 2^38 times a tap length <= sqrt(18): the range the self-test draws from); the reconstruction chain establishes that per tile (post.hip) and
// takes the plain `/` for a tile or wave that fails: the same result by definition.  cap_debug_get(CAP_DEBUG_SELFTEST_DIV) compares both
// forms on the device, bit for bit.  (The same treatment of sqrtf and of the small-scene shading's per-vertex divisions measured
// slower -- per-vertex guards are not amortised the way a tile's are: docs/experiments.md (70).)
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
}  // namespace cap
