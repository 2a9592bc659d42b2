// cap_math.h — fp32 arithmetic contract of the renderer, usable from host and gfx950 device code.
//
// The reference's shading code (shaders/*.h) leaves mad/dot/normalize/sin/cos/pow rounding to DXC and the
// driver.  This build pins them (DESIGN.md "fp32 arithmetic contract"): IEEE binary32, no implicit
// contraction (-ffp-contract=off), fma exactly where fmaf() is written, correctly rounded / and sqrt, and
// polynomial sin/cos/log2/exp2 made only of those operations — so host code, device code and the CPU
// oracle produce identical bits.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define CAP_HD __host__ __device__ __forceinline__

namespace cap
{
struct v3
{
    float x, y, z;
};

CAP_HD v3    mk3(float x, float y, float z) { return v3{x, y, z}; }
CAP_HD v3    operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
CAP_HD v3    operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
CAP_HD v3    operator*(v3 a, v3 b) { return v3{a.x * b.x, a.y * b.y, a.z * b.z}; }
CAP_HD v3    operator*(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
CAP_HD float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
CAP_HD v3    cross3(v3 a, v3 b)
{
    return v3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
CAP_HD v3 normalize3(v3 v)
{
    float inv = 1.0f / sqrtf(dot3(v, v));
    return v * inv;
}
CAP_HD float length3(v3 v) { return sqrtf(dot3(v, v)); }

CAP_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
CAP_HD float    u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

constexpr float kPi        = 3.141592653589793238463f;  // sampling.h:4
constexpr float kInvPi     = 1.0f / kPi;                // shading.h:16
constexpr float kRayEps    = 0.0001f;                   // lighting.h:45, rt_indirect.hlsl:156
constexpr float kRayFar    = 100000.0f;                 // lighting.h:31, shading.h:30
constexpr float kPrimaryFar = 1e6f;                     // camera.h:60

// Reciprocal of the intersection contract: integer seed + three Newton-Raphson steps, relative error <= 1e-7.
// Seven full-rate VALU operations instead of the ~11 (one of them quarter-rate) of a correctly rounded division, and,
// unlike v_rcp_f32, reproducible bit for bit on the host.
CAP_HD float rcp_c(float a)
{
    float x = u2f(0x7EF311C7u - f2u(a));
    x       = x * fmaf(-a, x, 2.0f);
    x       = x * fmaf(-a, x, 2.0f);
    x       = x * fmaf(-a, x, 2.0f);
    return x;
}

// sin/cos: Cody-Waite reduction by pi/2 (3 terms), Cephes sinf/cosf kernels on [-pi/4, pi/4].
CAP_HD void sincos_c(float x, float& s, float& c)
{
    const float kf = floorf(x * 0.636619772367581343f + 0.5f);
    const int   k  = (int)kf;
    float       a  = fmaf(-kf, 1.5703125f, x);
    a              = fmaf(-kf, 4.837512969970703125e-4f, a);
    a              = fmaf(-kf, 7.54978995489188216e-8f, a);
    const float z  = a * a;
    const float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    const float sp = fmaf(ps * z, a, a);
    const float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    const float cp = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    const bool  swap = (k & 1) != 0;
    const float s0 = swap ? cp : sp;
    const float c0 = swap ? sp : cp;
    s = (k & 2) ? -s0 : s0;
    c = ((k + 1) & 2) ? -c0 : c0;
}

CAP_HD float log2_c(float x)
{
    const uint32_t b = f2u(x);
    int            e = (int)((b >> 23) & 0xffu) - 127;
    float          m = u2f((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f)
    {
        m *= 0.5f;
        e += 1;
    }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    const float p = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 0.111111111111f, 0.142857142857f), 0.2f), 0.333333333333f), 1.0f);
    return fmaf((2.0f * s) * p, 1.44269504088896341f, (float)e);
}

CAP_HD float exp2_c(float y)
{
    const float n = floorf(y + 0.5f);
    const float f = y - n;
    float       p = 1.535336188319500e-4f;
    p             = fmaf(p, f, 1.339887440266574e-3f);
    p             = fmaf(p, f, 9.618437357674640e-3f);
    p             = fmaf(p, f, 5.550332471162809e-2f);
    p             = fmaf(p, f, 2.402264791363012e-1f);
    p             = fmaf(p, f, 6.931472028550421e-1f);
    p             = fmaf(p, f, 1.0f);
    return u2f(f2u(p) + ((uint32_t)(int)n << 23));
}

// scene.h:58  pow(kd, 2.2f) for kd in [0, 1]
CAP_HD float pow22_c(float x)
{
    if (!(x >= 1.17549435e-38f)) return 0.0f;
    const float y = 2.2f * log2_c(x);
    if (y < -125.0f) return 0.0f;
    return exp2_c(y);
}

// sampling.h:143-155 (evaluated on the host once per frame; the device reads the per-frame constant table)
inline void halton23(uint32_t frame_count, float& sx, float& sy)
{
    static const double pts[8][2] = {{0.5, 0.3333333333333333},   {0.25, 0.6666666666666666},
                                     {0.75, 0.1111111111111111},  {0.125, 0.4444444444444444},
                                     {0.625, 0.7777777777777777}, {0.375, 0.2222222222222222},
                                     {0.875, 0.5555555555555556}, {0.0625, 0.8888888888888888}};
    sx = (float)pts[frame_count % 8][0];
    sy = (float)pts[frame_count % 8][1];
}
}  // namespace cap
