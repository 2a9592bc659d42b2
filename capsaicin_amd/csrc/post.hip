// post.hip — gfx950 kernels of the reference's reconstruction chain, the stage right after the ray passes:
//   Gather -> Accumulate -> BlurDisocclusion -> Blur x2|x4 -> Combine -> TAA
// Reference (paths relative to /root/reference/src/core): shaders/spatial_gather.hlsl, temporal_accumulation.hlsl, eaw_blur.hlsl,
// combine_illumination.hlsl, eaw_edge_stopping.h, aabb.h, color_space.h, math_functions.h, utils.h, camera.h; pass order and
// buffer wiring src/systems/raytracing_system.cpp:262-317, 1283-1604, 1700-1790.  Full-resolution configuration
// (UPSCALE2X off, CALCULATE_VARIANCE / USE_VARIANCE on).  Buffers are fp32 float4 row-major images (the reference stores RGBA16F).
// Pure stencil / streaming work: HBM- and L2-bound, no MFMA.
#include "cap_kernels.h"
#include "cap_reproject.h"
#include "cap_unscaled.h"

#include <type_traits>

namespace cap
{
namespace
{
constexpr float kEpsPost = 1e-8f;  // math_functions.h:4

// length(float2(dx, dy)) of the 7 x 7 stencils' taps: sqrtf((float)(dx * dx + dy * dy)) evaluated once (correctly rounded square
// roots of small integers: the values the per-tap evaluation gives).  The loop indices are wave-uniform, so this is a scalar load.
struct TapLen7
{
    float v[7][7];
};
constexpr float csqrt(float x)
{
    // Newton iterations in double, then the correctly rounded float: x <= 18 is far from any rounding boundary issue
    double r = x > 0.f ? (double)x : 0.0;
    if (r == 0.0) return 0.f;
    double g = r;
    for (int i = 0; i < 40; ++i) g = 0.5 * (g + r / g);
    return (float)g;
}
constexpr TapLen7 make_len7()
{
    TapLen7 t{};
    for (int dy = -3; dy <= 3; ++dy)
        for (int dx = -3; dx <= 3; ++dx) t.v[dy + 3][dx + 3] = csqrt((float)(dx * dx + dy * dy));
    return t;
}
__constant__ TapLen7 kLen7 = make_len7();

__device__ __forceinline__ float    lerp1(float a, float b, float t) { return a + t * (b - a); }
__device__ __forceinline__ v3       div3(v3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float    luminance(v3 c) { return dot3(c, mk3(0.299f, 0.587f, 0.114f)); }

// exp(x), x <= 0 (pow(x, s), x in [0, 1]: pow01_t below): the exp2 / log2 polynomials of the arithmetic contract (cap_math.h)
__device__ __forceinline__ float exp_neg(float x)
{
    const float y = x * 1.44269504088896341f;
    if (!(y >= -125.0f)) return 0.0f;
    return exp2_c(y);
}
// math_functions.h:60-77
__device__ __forceinline__ float cubic(float x, float b, float c)
{
    float       y  = 0.0f;
    const float x2 = x * x, x3 = x * x * x;
    if (x < 1.0f)
        y = (12.0f - 9.0f * b - 6.0f * c) * x3 + (-18.0f + 12.0f * b + 6.0f * c) * x2 + (6.0f - 2.0f * b);
    else if (x <= 2.0f)
        y = (-b - 6.0f * c) * x3 + (6.0f * b + 30.0f * c) * x2 + (-12.0f * b - 48.0f * c) * x + (8.0f * b + 24.0f * c);
    return y / 6.0f;
}
// temporal_accumulation.hlsl:39-66.  The taps sit at whole-pixel offsets of the sample point, where cubic(1, 0, 0.5) is
// exactly 0: unless (c + 1) - c rounds away from 1, only the centre tap carries weight.  A tap whose kernel weight is
// exactly 0 adds +0 to both sums (history values are finite and non-negative), so it is skipped before its four loads.
// The kernel weight of tap (i, j) is cubic(|cur.x - c.x|) * cubic(|cur.y - c.y|) with cur = c + (i, j): three distinct values per
// axis.  They are evaluated once per sample point (BicubicTaps) instead of once per tap -- 6 cubics instead of 18 -- and shared by
// the images resampled at the same point (Accumulate's colour and moments histories: 6 instead of 36).  Same expressions on the
// same operands, so the same bits.
struct BicubicTaps
{
    f2    c;
    float kx[3], ky[3];
};
__device__ __forceinline__ BicubicTaps bicubic_taps(f2 uv, uint32_t w, uint32_t h)
{
    BicubicTaps t;
    t.c = uv_to_xy(uv, w, h);
#pragma unroll
    for (int i = -1; i <= 1; ++i)
    {
        const float cx = t.c.x + (float)i, cy = t.c.y + (float)i;
        t.kx[i + 1] = cubic(fabsf(cx - t.c.x), 0.0f, 0.5f);
        t.ky[i + 1] = cubic(fabsf(cy - t.c.y), 0.0f, 0.5f);
    }
    return t;
}
__device__ __forceinline__ v3 resample_bicubic(const Img& t, const BicubicTaps& b)
{
    v3       filtered = mk3(0.f, 0.f, 0.f);
    const f2 c        = b.c;
    float    tw       = 0.0f;
    for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j)
        {
            const f2 cur = f2{c.x + (float)i, c.y + (float)j};
            if (cur.x < 0.0f || cur.y < 0.0f || cur.x >= (float)t.w || cur.y >= (float)t.h) continue;
            const float kxy = b.kx[i + 1] * b.ky[j + 1];
            if (kxy == 0.0f) continue;
            const v3    value = sample_bilinear(t, xy_to_uv(cur, t.w, t.h));
            const float w     = kxy * (1.0f / (1.0f + luminance(value)));
            filtered = filtered + value * w;
            tw += w;
        }
    return tw > 1e-5f ? div3(filtered, tw) : mk3(0.f, 0.f, 0.f);
}
__device__ __forceinline__ v3 resample_bicubic(const Img& t, f2 uv) { return resample_bicubic(t, bicubic_taps(uv, t.w, t.h)); }
// math_functions.h:49-57
__device__ __forceinline__ v3 oct_decode(float fx, float fy)
{
    fx = fx * 2.0f - 1.0f, fy = fy * 2.0f - 1.0f;
    v3          n = mk3(fx, fy, 1.0f - fabsf(fx) - fabsf(fy));
    const float t = fminf(fmaxf(-n.z, 0.0f), 1.0f);
    n.x += n.x >= 0.0f ? -t : t;
    n.y += n.y >= 0.0f ? -t : t;
    return normalize3(n);
}
// ---- IEEE division without its scaling steps (exact mode, round 5) ----
// hipcc expands a / b into v_div_scale x 2, v_rcp, seven FMAs, v_div_fmas and v_div_fixup: ~47 SIMD cycles, four times per tap of
// the exact stencils.  The two v_div_scale only act when an exponent is extreme -- a denormal or huge denominator, a numerator below
// 2^-103, a quotient that would be denormal or whose exponents differ by 96 or more (CDNA ISA, V_DIV_SCALE_F32) --, and
// v_div_fixup only on zeros, infinities and NaNs; everywhere else they pass their operands through and what remains is the
// sequence below: the same instructions on the same operands, so the same bits (29 cycles).  `div_unscaled` is therefore ONLY called
// where the operands are known to be in range -- |a| = 0 or in [2^-80, 2^41], b in [2^-40, 2^41) -- which the stencil kernels
// establish per tile while staging (depths in [1e-5, 2^40], luminances 0 or in [2^-50, 2^30], the pixel's sigmas in [2^-38, 2^38];
// a tile or wave that fails takes the IEEE form: the same result by definition).  cap_debug_get(CAP_DEBUG_SELFTEST_DIV) compares both
// forms on the device: every float for log2's (m - 1) / (m + 1), 2^30 pseudo-random pairs over the whole stated range for the rest.
template <bool NS>
__device__ __forceinline__ float div_c(float a, float b)
{
    return NS ? div_unscaled(a, b) : a / b;
}
constexpr float kDivHi = 1.0995116e12f /* 2^40 */;
__device__ __forceinline__ bool div_sigma_ok(float s) { return s == 0.0f || (s >= 3.637979e-12f /* 2^-38 */ && s <= 2.7487791e11f /* 2^38 */); }
__device__ __forceinline__ bool div_depth_ok(float d) { return d < 1e-5f || d <= kDivHi; }  // (NaN: both compares false)
__device__ __forceinline__ bool div_luma_ok(float l) { return l == 0.0f || (l >= 8.8817842e-16f /* 2^-50 */ && l <= 1.0737418e9f /* 2^30 */); }

// cap_math.h log2_c with the unscaled division: m in (0.7071, 1.4143], so (m - 1) in [-0.293, 0.415] is 0 or at least 2^-24 in
// magnitude and (m + 1) in [1.7, 2.42] -- never scaled, whatever x (compared for every float by the self-test)
__device__ __forceinline__ float log2_c_ns(float x)
{
    const uint32_t b = f2u(x);
    int            e = (int)((b >> 23) & 0xffu) - 127;
    float          m = u2f((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f)
    {
        m *= 0.5f;
        e += 1;
    }
    const float s = div_unscaled(m - 1.0f, m + 1.0f);
    const float z = s * s;
    const float p = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 0.111111111111f, 0.142857142857f), 0.2f), 0.333333333333f), 1.0f);
    return fmaf((2.0f * s) * p, 1.44269504088896341f, (float)e);
}
template <bool NS>
__device__ __forceinline__ float pow01_t(float x, float s)
{
    if (!(x >= 1.17549435e-38f)) return 0.0f;
    const float y = s * (NS ? log2_c_ns(x) : log2_c(x));
    if (!(y >= -125.0f)) return 0.0f;
    return exp2_c(y);
}

// eaw_edge_stopping.h (NS: divisions in the unscaled form, see div_unscaled)
template <bool NS = false>
__device__ __forceinline__ float normal_weight(v3 n0, v3 n1, float s) { return pow01_t<NS>(fmaxf(dot3(n0, n1), 0.0f), s); }
// (NS: the caller has established s_depth > 0, so sigma = s_depth * length(tap) is 0 exactly at the centre tap)
template <bool NS = false>
__device__ __forceinline__ float depth_weight(float dc, float dp, float s, bool centre = false)
{
    const float t = NS ? (centre ? 0.0f : div_unscaled(fabsf(dc - dp), s)) : (s == 0.0f ? 0.0f : fabsf(dc - dp) / s);
    return exp_neg(-t);
}
template <bool NS = false>
__device__ __forceinline__ float luma_weight(float lc, float lp, float s) { return exp_neg(-div_c<NS>(fabsf(lc - lp), s)); }

// CapPostSettings::fast_weights: the same three weights through the hardware's transcendental instructions -- v_log_f32 / v_exp_f32
// (base 2, 1 ulp) and v_rcp_f32 instead of the contract's polynomials and IEEE divisions: ~12 instructions where the exact forms
// take ~110.  Not bit-comparable with the oracle; held to the tolerance stated in the header (tests/test_post_gpu.py).  The
// per-pixel reciprocals are hoisted: `neg_inv_s` = -log2(e) / sigma, or 0 where the reference's sigma is 0 (weight 1).
constexpr float kLog2e = 1.44269504088896341f;
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_neg_inv(float s) { return s == 0.0f ? 0.0f : -kLog2e * fast_rcp(s); }
__device__ __forceinline__ float fast_normal_weight(v3 n0, v3 n1, float s)
{
    // log2(0) = -inf, s * -inf = -inf, exp2(-inf) = 0: the exact form's pow(0, s) = 0 without a branch
    return __builtin_amdgcn_exp2f(s * __builtin_amdgcn_logf(fmaxf(dot3(n0, n1), 0.0f)));
}
__device__ __forceinline__ float fast_exp_weight(float a, float b, float neg_inv_s) { return __builtin_amdgcn_exp2f(fabsf(a - b) * neg_inv_s); }
// 1 / length(float2(dx, dy)) of the 7 x 7 taps, 0 for the centre (the reference's sigma * 0: weight 1)
struct TapInv7
{
    float v[7][7];
};
constexpr TapInv7 make_inv7()
{
    TapInv7 t{};
    for (int dy = -3; dy <= 3; ++dy)
        for (int dx = -3; dx <= 3; ++dx) t.v[dy + 3][dx + 3] = (dx || dy) ? 1.0f / csqrt((float)(dx * dx + dy * dy)) : 0.0f;
    return t;
}
__constant__ TapInv7 kInv7 = make_inv7();
// color_space.h
__device__ __forceinline__ v3 rgb2ycocg(v3 c)
{
    return mk3(c.x / 4.0f + c.y / 2.0f + c.z / 4.0f, c.x / 2.0f - c.z / 2.0f, -c.x / 4.0f + c.y / 2.0f - c.z / 4.0f);
}
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ v3    ycocg2rgb(v3 c) { return mk3(clamp01(c.x + c.y - c.z), clamp01(c.x + c.z), clamp01(c.x - c.y - c.z)); }
__device__ __forceinline__ v3    simple_tonemap(v3 v) { return div3(v, 1.0f + luminance(v)); }
__device__ __forceinline__ v3    invert_simple_tonemap(v3 v) { return div3(v, 1.0f - luminance(v)); }

// camera.h:64-80
__device__ __forceinline__ v3 reconstruct_world_position(const CameraDev& cam, f2 uv, float depth)
{
    const float cx = (uv.x - 0.5f) * cam.sensor_x, cy = (uv.y - 0.5f) * cam.sensor_y;
    const v3    d  = normalize3(mk3(fmaf(cy, cam.up[0], fmaf(cx, cam.right[0], cam.focal_length * cam.forward[0])),
                                    fmaf(cy, cam.up[1], fmaf(cx, cam.right[1], cam.focal_length * cam.forward[1])),
                                    fmaf(cy, cam.up[2], fmaf(cx, cam.right[2], cam.focal_length * cam.forward[2]))));
    return mk3(cam.position[0], cam.position[1], cam.position[2]) + d * depth;
}

// 32 x 8 pixel workgroups: a wave covers two 32-pixel row segments (coalesced 512-B rows)
__device__ __forceinline__ bool pixel_of_thread(uint32_t w, uint32_t h, uint32_t& x, uint32_t& y)
{
    x = blockIdx.x * 32u + (threadIdx.x & 31u);
    y = blockIdx.y * 8u + (threadIdx.x >> 5);
    return x < w && y < h;
}

// The three stencil filters decode every neighbour's octahedral normal (a normalize = sqrt + division per tap, ~200 taps per
// pixel and frame over the whole chain).  The decode depends on the texel only, so it is done once per pixel and frame here:
// out = (OctDecode(nd.xy), nd.w) — the same values the per-tap decode would give, a quarter of the filters' VALU work less.
__global__ __launch_bounds__(kBlock) void k_decode_normals(const float4* nd, float4* out, uint32_t n)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    {
        const float4 g = nd[i];
        const v3     d = oct_decode(g.x, g.y);
        out[i]         = make_float4(d.x, d.y, d.z, g.w);
    }
}

// PostChainArgs::tiled: the render's planes are tile-ordered (cap_device.h ScreenDev: 8 x 8 tiles, row-major tile grid, one shard);
// index of pixel (x, y) in such a plane, or its row-major index when tiles_x is 0
__device__ __forceinline__ size_t aux_index(uint32_t x, uint32_t y, uint32_t w, uint32_t tiles_x)
{
    return tiles_x ? (size_t)((y >> 3) * tiles_x + (x >> 3)) * kTilePixels + ((y & 7u) << 3 | (x & 7u)) : (size_t)y * w + x;
}
// tile-ordered indirect and normal/depth planes -> row-major indirect image and decoded (normal.xyz, depth) image: the untiling the
// chain needs for its stencils and k_decode_normals in one pass (direct and albedo are only read pointwise: Combine takes them tiled)
__global__ __launch_bounds__(kBlock) void k_untile_decode(ScreenDev sc, const float4* color, const float4* nd, float4* out_color, float4* out_normals)
{
    const uint32_t n = sc.local_tiles * kTilePixels;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < n; pl += gridDim.x * kBlock)
    {
        uint32_t x, y;
        if (!local_pixel_to_xy(sc, pl, x, y)) continue;
        const size_t o = (size_t)y * sc.width + x;
        if (color) out_color[o] = color[pl];
        const float4 g = nd[pl];
        const v3     d = oct_decode(g.x, g.y);
        out_normals[o] = make_float4(d.x, d.y, d.z, g.w);
    }
}

// spatial_gather.hlsl:28-109 in its UPSCALE2X form (the full-resolution form is k_stencil_lds<kGather> below).  nd = decoded
// (normal.xyz, depth) image.  UP (UPSCALE2X, :36-46, :83-87): the grid and `color` are half resolution and the G-buffer is read at (xy << 1) + (ox, oy).
// The taps are bounded by the FULL window size, as the host passes it (raytracing_system.cpp:1562-1569): a tap beyond the
// half-resolution image reads a G-buffer texel outside the window, i.e. depth 0, and is skipped as background.
template <bool UP, bool FAST = false>
__global__ __launch_bounds__(kBlock) void k_gather(PostSettingsDev s, Img color, Img nd, float4* out, int ox, int oy)
{
    uint32_t x, y;
    if (!pixel_of_thread(color.w, color.h, x, y)) return;
    const int    bound_w = UP ? (int)nd.w : (int)color.w, bound_h = UP ? (int)nd.h : (int)color.h;
    const float4 cg = UP ? ldi(nd, ((int)x << 1) + ox, ((int)y << 1) + oy) : ld(nd, x, y);
    const v3     cn = xyz(cg);
    const float  cd = cg.w;
    const v3     cc = xyz(ld(color, x, y));
    float4       res;
    if (cd < 1e-5f)
        res = make_float4(cc.x, cc.y, cc.z, 0.0f);
    else
    {
        const float s_depth = cd * s.gather_depth_sigma, s_normal = s.gather_normal_sigma, s_luma = s.gather_luma_sigma;
        const float f_depth = FAST ? fast_neg_inv(s_depth) : 0.f, f_luma = FAST ? fast_neg_inv(s_luma) : 0.f, lcc = luminance(cc);
        v3          filtered = mk3(0.f, 0.f, 0.f);
        float       total    = 0.0f;
        for (int dy = -3; dy <= 3; ++dy)
            for (int dx = -3; dx <= 3; ++dx)
            {
                const int sx = (int)x + dx, sy = (int)y + dy;
                if (sx < 0 || sy < 0 || sx >= bound_w || sy >= bound_h) continue;
                const v3     c = xyz(ldi(color, sx, sy));
                const float4 g = UP ? ldi(nd, (sx << 1) + ox, (sy << 1) + oy) : ldi(nd, sx, sy);
                if (g.w < 1e-5f) continue;
                const v3    n   = xyz(g);
                float       wgt;
                if (FAST)
                    wgt = fast_normal_weight(cn, n, s_normal) * fast_exp_weight(cd, g.w, f_depth * kInv7.v[dy + 3][dx + 3]) * fast_exp_weight(lcc, luminance(c), f_luma);
                else
                    wgt = normal_weight(cn, n, s_normal) * depth_weight(cd, g.w, s_depth * kLen7.v[dy + 3][dx + 3]) * luma_weight(lcc, luminance(c), s_luma);
                filtered = filtered + c * wgt;
                total += wgt;
            }
        const v3 r = (total < kEpsPost) ? cc : (FAST ? filtered * fast_rcp(total) : div3(filtered, total));
        res        = make_float4(r.x, r.y, r.z, 1.0f);
    }
    out[(size_t)y * color.w + x] = res;
}

// out[(y, x)] = full[(2y + oy, 2x + ox)]
__global__ __launch_bounds__(kBlock) void k_decimate2x(const float4* full, uint32_t w, uint32_t h, uint32_t ox, uint32_t oy, float4* out)
{
    const uint32_t w2 = w >> 1, h2 = h >> 1;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < w2 * h2; i += gridDim.x * kBlock)
    {
        const uint32_t y = i / w2, x = i - y * w2;
        out[i] = full[(size_t)(2u * y + oy) * w + (2u * x + ox)];
    }
}

// temporal_accumulation.hlsl:179-205
__device__ __forceinline__ float closest_depth(const Img& g, f2 xy)
{
    float closest = ldi(g, (int)xy.x, (int)xy.y).w;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
        {
            const int tx = (int)xy.x + dx, ty = (int)xy.y + dy;
            if ((float)tx >= (float)g.w || (float)ty >= (float)g.h || tx < 0 || ty < 0) continue;
            const float4 v = ldi(g, tx, ty);
            if (v.w != 0.0f && v.w < closest) closest = v.w;
        }
    return closest;
}

// One bilinear tap with the hardware's arithmetic for the FAST variants below: the reference's footprint and weights (utils.h:20-35),
// no IEEE division.  (uv -> pixel coordinates as the reference does it, so the four texels are the ones the exact code blends.)
__device__ __forceinline__ v3 sample_bilinear_fast(const Img& t, f2 uv)
{
    const f2       xy = uv_to_xy(uv, t.w, t.h);
    const float    fx = xy.x - 0.5f, fy = xy.y - 0.5f;
    const float    flx = floorf(fx), fly = floorf(fy);
    const uint32_t ux = sat_uint(flx), uy = sat_uint(fly);
    const float    wx = fx - flx, wy = fy - fly;
    const v3 v00 = xyz(ld(t, ux, uy)), v01 = xyz(ld(t, ux, uy + 1)), v10 = xyz(ld(t, ux + 1, uy)), v11 = xyz(ld(t, ux + 1, uy + 1));
    return lerp3(lerp3(v00, v10, wx), lerp3(v01, v11, wx), wy);
}

// temporal_accumulation.hlsl:213-325.  With UPSCALE2X (s.lowres_indirect) `color` is the half-resolution image (SampleColor uses its
// size, :228-235) and a pixel that got no new sample this frame keeps its history (:307-313).
// FAST (CapPostSettings::fast_weights): everything that DECIDES -- the reprojection, the disocclusion test, the history length -- is
// the exact code on the same G-buffer and cameras, so the fast chain resets and blends exactly where the exact one does; only the
// values are cheaper: the colour sample at a pixel centre is the texel (the reference's bilinear tap there has fractional weights of
// 0 or one rounding error), and the bicubic history resample is its centre tap (the other eight sit where the Mitchell weight is
// exactly 0 unless (c + 1) - c rounds: see resample_bicubic), without the weight that cancels in filtered / total.
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_accumulate(PostSettingsDev s, uint32_t frame_count, CameraDev cam, CameraDev prev_cam, Img color,
                                                       Img nd, Img color_history, Img moments_history, Img prev_nd, float4* out_color,
                                                       float4* out_moments)
{
    uint32_t x, y;
    const uint32_t W = nd.w, H = nd.h;
    if (!pixel_of_thread(W, H, x, y)) return;
    const f2     uv = f2{((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
    const float4 g  = ld(nd, x, y);
    const size_t o  = (size_t)y * W + x;
    if (FAST && color.w == W && color.h == H)
    {
        // The same decisions as below, but every load of the pixel -- the 3 x 3 depths around the reprojected position, the history
        // length, the two histories' bilinear footprints, the colour -- is issued in ONE batch behind the reprojection, with clamped
        // coordinates where the exact code would not load at all, and the branches become selects at the end: the exact code's four
        // dependent round trips (G-buffer -> previous depths -> history length -> histories) were what this kernel waited for.
        const bool  bg  = g.w < 1e-5f;
        const v3    hit = reconstruct_world_position(cam, uv, bg ? 1.0f : g.w);
        const f2    puv = image_plane_uv(prev_cam, hit);
        const bool  off = puv.x < 0.0f || puv.y < 0.0f || puv.x > 1.0f || puv.y > 1.0f || frame_count == 0;
        const f2    sp  = (bg || off) ? f2{0.5f, 0.5f} : puv;  // where nothing is reprojected: any in-image position, result unused
        const f2    pxy = uv_to_xy(sp, W, H);
        const int   cx = (int)pxy.x, cy = (int)pxy.y;
        float       dn[9];
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
            {
                const int  tx = cx + dx, ty = cy + dy;
                const bool in = tx >= 0 && ty >= 0 && tx < (int)W && ty < (int)H;
                const float d = prev_nd.p[(size_t)min(max(ty, 0), (int)H - 1) * W + min(max(tx, 0), (int)W - 1)].w;
                dn[(dx + 1) * 3 + dy + 1] = in ? d : 0.0f;  // (a depth of 0 is skipped by the minimum below, like an out-of-bounds tap)
            }
        const float hl  = ld(moments_history, sat_uint(floorf(pxy.x)), sat_uint(floorf(pxy.y))).w;
        const f2    c0  = uv_to_xy(sp, W, H);
        const f2    cuv = f2{fminf(fmaxf(c0.x * fast_rcp((float)W), 0.0f), 1.0f), fminf(fmaxf(c0.y * fast_rcp((float)H), 0.0f), 1.0f)};
        const v3    history = sample_bilinear_fast(color_history, cuv), mh = sample_bilinear_fast(moments_history, cuv);
        // (the last row and column are NOT the texel: UVtoXY clamps to dim - 1, utils.h:6-10, so the reference's tap there is a two-texel blend)
        const v3    c = (x + 1u < W && y + 1u < H) ? xyz(color.p[o]) : sample_bilinear(color, uv);
        const float l = luminance(c);
        // closest_depth (:179-205): the centre texel, then the minimum over the non-zero depths of the 3 x 3
        float closest = dn[4];
#pragma unroll
        for (int k = 0; k < 9; ++k)
            if (dn[k] != 0.0f && dn[k] < closest) closest = dn[k];
        const float cur_depth = length3(hit - mk3(prev_cam.position[0], prev_cam.position[1], prev_cam.position[2]));
        const bool  reset     = bg || off || fabsf(closest - cur_depth) / cur_depth > 0.05f;
        float       alpha     = s.temporal_upscale_feedback;
        uint32_t    hist_len  = sat_uint(hl);
        if (hist_len < 256u) alpha = fminf(alpha, 1.0f - 1.0f / (float)(hist_len + 1));
        const float m0 = lerp1(l, mh.x, alpha), m1 = lerp1(l * l, mh.y, alpha);
        const v3    blended = lerp3(c, history, alpha);
        out_moments[o] = reset ? make_float4(l, l * l, 0.0f, 1.0f) : make_float4(m0, m1, 0.0f, (float)(hist_len + 1));
        out_color[o]   = reset ? make_float4(c.x, c.y, c.z, 0.0f) : make_float4(blended.x, blended.y, blended.z, fabsf(m1 - m0 * m0));
        return;
    }
    bool         reset = g.w < 1e-5f;
    f2           puv = f2{0.f, 0.f}, pxy = f2{0.f, 0.f};
    if (!reset)
    {
        const v3 hit = reconstruct_world_position(cam, uv, g.w);
        puv          = image_plane_uv(prev_cam, hit);
        reset        = puv.x < 0.0f || puv.y < 0.0f || puv.x > 1.0f || puv.y > 1.0f || frame_count == 0;
        if (!reset)
        {
            pxy = uv_to_xy(puv, W, H);
            const float cur_depth  = length3(hit - mk3(prev_cam.position[0], prev_cam.position[1], prev_cam.position[2]));
            const float prev_depth = closest_depth(prev_nd, pxy);
            reset                  = fabsf(prev_depth - cur_depth) / cur_depth > 0.05f;
        }
    }
    const v3    c = sample_bilinear(color, uv);
    const float l = luminance(c);
    if (reset)
    {
        out_color[o]   = make_float4(c.x, c.y, c.z, 0.0f);
        out_moments[o] = make_float4(l, l * l, 0.0f, 1.0f);
        return;
    }
    float    alpha    = s.temporal_upscale_feedback;
    uint32_t hist_len = sat_uint(ld(moments_history, sat_uint(floorf(pxy.x)), sat_uint(floorf(pxy.y))).w);
    if (hist_len < 256u)
    {
        const float t = 1.0f / (float)(hist_len + 1);
        alpha         = fminf(alpha, 1.0f - t);
    }
    if (s.lowres_indirect && ((x % 2u) != (frame_count % 4u) / 2u || (y % 2u) != (frame_count % 4u) % 2u))
    {
        alpha = 1.0f;
        hist_len -= 1u;  // uint: a length of 0 wraps and the + 1 below brings it back to 0
    }
    const BicubicTaps taps    = bicubic_taps(puv, color_history.w, color_history.h);  // both histories are W x H images
    const v3          history = resample_bicubic(color_history, taps), mh = resample_bicubic(moments_history, taps);
    const float m0 = lerp1(l, mh.x, alpha), m1 = lerp1(l * l, mh.y, alpha);
    const float variance = fabsf(m1 - m0 * m0);
    out_moments[o]   = make_float4(m0, m1, 0.0f, (float)(hist_len + 1));
    const v3 blended = lerp3(c, history, alpha);
    out_color[o]     = make_float4(blended.x, blended.y, blended.z, variance);
}

__device__ __forceinline__ v3 remove_fireflies(float4 v) { return mk3(fminf(v.x, 10.0f), fminf(v.y, 10.0f), fminf(v.z, 10.0f)); }

// ------------------------------------------------------------------------------------------------
// The three stencil filters -- Gather (spatial_gather.hlsl:28-109, 7 x 7), BlurDisocclusion (eaw_blur.hlsl:142-223, 7 x 7) and the
// a-trous Blur passes (eaw_blur.hlsl:48-137, 5 x 5 at strides 1, 3, 5, 7) -- as LDS-tiled kernels, in BOTH weight modes (round 5;
// until round 4 the exact mode read every tap from global memory behind two skip branches).
//   * A workgroup stages its footprint once -- (colour | variance) and (decoded normal | depth) as two float4 tiles -- and the taps
//     are ds_read_b128 at compile-time offsets; the XCD-aware tile order keeps neighbouring tiles' halos in one L2.
//   * No branches in the tap loop.  A tap outside the image or on the background gets weight 0 by a select, and its texel is staged
//     with colour 0 (and moments 0): the skipped tap of the reference then adds c * 0 = +0 to sums of non-negative terms, which
//     leaves every sum's bits as they are -- so the exact mode stays bit-identical to the oracle's `continue` (tests/test_post_gpu.py).
//   * Exact mode (FAST = false): the three weights in the contract's arithmetic, multiplied in the reference's order.
//     fast_weights (FAST = true): the product of the three weights as ONE exponential, 2^(s_n log2(n.n') - |d - d'| / s_d - |l - l'| / s_l)
//     -- one v_log_f32 and one v_exp_f32 per tap -- and v_rcp_f32 for the final divisions; toleranced (header, tests).
//   * Strides 3 / 5 / 7: a pixel's 25 taps all lie in rows congruent to its own modulo the stride, so a workgroup takes 8 rows of ONE
//     row phase (y = phase + stride * j) and 64 contiguous columns: a footprint of (64 + 4 stride) x 12 texels = 29 / 32 / 35 KB of LDS
//     -- four 512-thread workgroups per CU, staging and arithmetic of different workgroups overlap -- where round 3's square 64 x 16 tile
//     with its 2-stride halo on every side took 68 / 97 / 129.5 KB and ran one workgroup per CU at strides 5 and 7 (2.2 staged texels
//     per pixel instead of 4.0 at stride 7).
// ------------------------------------------------------------------------------------------------
// XCD-aware workgroup -> tile mapping.  Workgroups are dealt round-robin to the eight XCDs, so with the plain row-major mapping
// horizontally adjacent tiles -- whose halos overlap -- sit on different XCDs and every XCD's L2 fetches its own copy of the shared
// texels.  Here XCD k (workgroup id mod 8) takes the k-th eighth of the tiles in linear order: a band of adjacent tile rows, whose
// halos meet in one L2.  (Speed only: the mapping is a bijection whatever the placement.)  The grid has 8 * ceil(tiles / 8)
// workgroups (xcd_grid); the few whose tile index falls off the end leave at once (returns false).
__device__ __forceinline__ bool xcd_tile_index(uint32_t n, uint32_t& t)
{
    const uint32_t per = (n + 7u) / 8u, id = blockIdx.x;
    t = (id % 8u) * per + id / 8u;
    return t < n;
}
__device__ __forceinline__ bool xcd_tile(uint32_t tiles_x, uint32_t tiles_y, uint32_t& bx, uint32_t& by)
{
    uint32_t t;
    if (!xcd_tile_index(tiles_x * tiles_y, t)) return false;
    by = t / tiles_x, bx = t - by * tiles_x;
    return true;
}
inline uint32_t xcd_grid(uint32_t tiles) { return 8u * ((tiles + 7u) / 8u); }

// kernel_weights[abs(d)] of eaw_blur.hlsl:50: {1, 2/3, 1/6} (selects, so that a loop index that is not unrolled stays a scalar)
__device__ __forceinline__ float atrous_kernel(int d)
{
    d = d < 0 ? -d : d;
    return d == 0 ? 1.0f : (d == 1 ? 2.0f / 3.0f : 1.0f / 6.0f);
}

enum StencilKind
{
    kGather = 0,
    kDisocclusion,
    kBlur
};

struct TapCenter
{
    v3    cn, cc;
    float cd, lcc;
    float s_normal, s_depth, s_luma;  // exact mode: the sigmas as the reference forms them
    float f_depth, f_luma;            // fast mode: -log2(e) / sigma (0 where the reference's sigma is 0: weight 1)
};

// filtered += c * k.  Exact mode: a product and a sum, as the oracle rounds them (-ffp-contract=off).  Fast mode: three fused multiply-adds
// (the compiler may not contract, so the toleranced mode writes them out: 4 of a tap's ~27 instructions less).
template <bool FAST>
__device__ __forceinline__ v3 acc3(v3 f, v3 c, float k)
{
    return FAST ? mk3(fmaf(c.x, k, f.x), fmaf(c.y, k, f.y), fmaf(c.z, k, f.z)) : f + c * k;
}

// Weights of one tap.  Exact: wgt = normal * depth (the reference's first two factors, in its order) and lw = the luminance weight
// (1 when the pass has none).  Fast: wgt = the whole product from one exponential, lw = 1.
// len / inv_len: length(float2(dx, dy)) of the tap and its reciprocal (0 for the centre: the reference's sigma * 0 gives weight 1).
template <bool FAST, bool LUMA, bool NS = false>
__device__ __forceinline__ void tap_weights(const TapCenter& k, float4 g, v3 c, float len, float inv_len, float& wgt, float& lw, bool centre = false)
{
    if (FAST)
    {
        float e = k.s_normal * __builtin_amdgcn_logf(fmaxf(dot3(k.cn, xyz(g)), 0.0f));  // log2(0) = -inf carries a normal weight of 0
        e       = fmaf(fabsf(k.cd - g.w), k.f_depth * inv_len, e);
        if (LUMA) e = fmaf(fabsf(k.lcc - luminance(c)), k.f_luma, e);
        wgt = __builtin_amdgcn_exp2f(e), lw = 1.0f;
    }
    else
    {
        wgt = normal_weight<NS>(k.cn, xyz(g), k.s_normal) * depth_weight<NS>(k.cd, g.w, k.s_depth * len, centre);
        lw  = LUMA ? luma_weight<NS>(k.lcc, luminance(c), k.s_luma) : 1.0f;
    }
}

// One texel of a stencil footprint for the LDS tiles: outside the image -> depth 0 (= background, never taken) and colour 0; a
// background texel inside the image keeps its (normal | depth) and gets colour 0 -- a tap of weight 0 still multiplies what it
// reads, so EVERY plane of an entry that can be a skipped tap is a defined, finite 0.
// TILED (Gather on an unsharded context): `color` and `nd` are the render's tile-ordered planes, `nd` still octahedral-encoded -- the
// texel is untiled and its normal decoded here, with the operations of k_untile_decode (the same bits), so that pass and its image
// round trip are not needed.
// DUAL (the first a-trous pass): `color` is IntegrateTemporally's output and `alt` BlurDisocclusion's, which is only written where a
// pixel needed its 7 x 7 filter; a pixel that BlurDisocclusion passes through (background, or a history of eight frames: most of
// the image most of the time) is taken from `color` as that pass would have written it -- its fireflies clamped, which CLAMP does
// here anyway, and its variance kept -- so in the steady state BlurDisocclusion neither reads nor writes an image.
// `ok` (exact mode): the texel's depth and luminance are in the range div_unscaled is used on.
template <bool CLAMP, bool MOMENTS, bool TILED = false, bool DUAL = false>
__device__ __forceinline__ void stage_texel(const Img& color, const Img& nd, const Img& moments, int sx, int sy, float4& c, float4& g, float2& mm,
                                            uint32_t tiles_x = 0, const float4* alt = nullptr, bool* ok = nullptr)
{
    const bool   in = sx >= 0 && sy >= 0 && sx < (int)color.w && sy < (int)color.h;
    // (clamped address: the loads are unconditional, their values selected)
    const size_t o  = !in ? 0 : (TILED ? aux_index((uint32_t)sx, (uint32_t)sy, color.w, tiles_x) : (size_t)sy * color.w + sx);
    float4       gv = nd.p[o];
    float4       v  = color.p[o];
    if (DUAL)
    {
        const float hist = moments.p[o].w;
        if (!(gv.w < 1e-5f || hist >= 8.0f)) v = alt[o];
    }
    if (TILED)
    {
        const v3 d = oct_decode(gv.x, gv.y);
        gv         = make_float4(d.x, d.y, d.z, gv.w);
    }
    float        m0 = 0.f, m1 = 0.f;
    if (MOMENTS)
    {
        const float4 t = moments.p[o];
        m0 = t.x, m1 = t.y;
    }
    const bool take = in && !(gv.w < 1e-5f);
    const float cx = CLAMP ? fminf(v.x, 10.0f) : v.x, cy = CLAMP ? fminf(v.y, 10.0f) : v.y, cz = CLAMP ? fminf(v.z, 10.0f) : v.z;  // remove_fireflies
    c  = make_float4(take ? cx : 0.f, take ? cy : 0.f, take ? cz : 0.f, take ? v.w : 0.f);
    g  = make_float4(in ? gv.x : 0.f, in ? gv.y : 0.f, in ? gv.z : 0.f, in ? gv.w : 0.f);
    mm = make_float2(take ? m0 : 0.f, take ? m1 : 0.f);
    if (ok) *ok = div_depth_ok(g.w) && div_luma_ok(luminance(xyz(c)));
}

// Stride-1 stencils on a 32 x 8 pixel tile + halo R: Gather (R = 3), BlurDisocclusion (R = 3), the first a-trous pass (R = 2).
// USE_VAR = the USE_VARIANCE define of eaw_blur.hlsl (:68, :114, :127, :162-165).
// TILED (Gather only): see stage_texel; the decoded (normal | depth) of the workgroup's own pixels goes to `normals_out` row-major --
// the image every later pass reads the G-buffer through.
// DUAL (the first a-trous pass only): see stage_texel; `alt` = BlurDisocclusion's output, `moments` = the history lengths.
template <int KIND, int R, bool USE_VAR, bool FAST, bool TILED = false, bool DUAL = false>
__global__ __launch_bounds__(kBlock) void k_stencil_lds(PostSettingsDev s, Img color, Img nd, Img moments, float4* out, uint32_t tiles_x = 0,
                                                        float4* normals_out = nullptr, const float4* alt = nullptr)
{
    static_assert(!TILED || KIND == kGather, "only Gather reads the render's planes");
    static_assert(!DUAL || KIND == kBlur, "only the first a-trous pass reads BlurDisocclusion's sparse output");
    constexpr int TW = 32 + 2 * R, TH = 8 + 2 * R;
    __shared__ float4 t_col[TW * TH], t_nd[TW * TH];
    __shared__ float2 t_mom[KIND == kDisocclusion ? TW * TH : 1];
    const int W = (int)color.w;
    uint32_t  bx, by;
    if (!xcd_tile((color.w + 31u) / 32u, (color.h + 7u) / 8u, bx, by)) return;  // workgroup-uniform
    const int      x0 = (int)(bx * 32u) - R, y0 = (int)(by * 8u) - R;
    const uint32_t x = bx * 32u + (threadIdx.x & 31u), y = by * 8u + (threadIdx.x >> 5);
    const bool     in_image = x < color.w && y < color.h;
    const size_t   o        = (size_t)y * W + x;
    // the centre pixel comes from global memory: a background centre passes its own colour through, which the tile does not hold
    float4 cv = make_float4(0.f, 0.f, 0.f, 0.f), cg = cv;
    float  hist = 0.0f;
    if (in_image)
    {
        if (TILED)
        {
            const size_t ot = aux_index(x, y, color.w, tiles_x);
            cv = color.p[ot], cg = nd.p[ot];
            const v3 d = oct_decode(cg.x, cg.y);
            cg         = make_float4(d.x, d.y, d.z, cg.w);
            normals_out[o] = cg;
        }
        else
            cv = color.p[o], cg = nd.p[o];
        if (KIND == kDisocclusion) hist = moments.p[o].w;
        if (DUAL)
            if (!(cg.w < 1e-5f || moments.p[o].w >= 8.0f)) cv = alt[o];
    }
    const bool pass = cg.w < 1e-5f || (KIND == kDisocclusion && hist >= 8.0f);
    // BlurDisocclusion passes most pixels through once the history is eight frames long: a workgroup of such pixels stages nothing
    bool needs_taps = true;
    if (KIND == kDisocclusion)
    {
        needs_taps = __syncthreads_or((in_image && !pass) ? 1 : 0) != 0;
        if (!needs_taps) return;  // every pixel of the tile passes through: the first a-trous pass reads them where they are (DUAL)
    }
    bool tile_ok = true;  // exact mode: every staged depth and luminance is in div_unscaled's range (workgroup-uniform)
    if (needs_taps)
    {
        bool bad = false;
        for (int e = (int)threadIdx.x; e < TW * TH; e += (int)kBlock)
        {
            float4 c, g;
            float2 mm;
            bool   ok = true;
            stage_texel<KIND != kGather, KIND == kDisocclusion, TILED, DUAL>(color, nd, moments, x0 + e % TW, y0 + e / TW, c, g, mm, tiles_x, alt,
                                                                             FAST ? nullptr : &ok);
            bad |= !ok;
            t_col[e] = c, t_nd[e] = g;
            if (KIND == kDisocclusion) t_mom[e] = mm;
        }
        if (FAST)
            __syncthreads();
        else
            tile_ok = __syncthreads_or(bad ? 1 : 0) == 0;
    }
    if (!in_image) return;
    TapCenter k;
    k.cn = xyz(cg), k.cd = cg.w;
    k.cc = KIND == kGather ? xyz(cv) : remove_fireflies(cv), k.lcc = luminance(k.cc);
    const float cvar = (KIND == kGather) ? 0.0f : (USE_VAR ? cv.w : 0.0f);
    float4      res  = make_float4(k.cc.x, k.cc.y, k.cc.z, cvar);
    if (KIND == kGather) res.w = pass ? 0.0f : 1.0f;
    if (!pass)
    {
        const float sd = KIND == kGather ? s.gather_depth_sigma : s.eaw_depth_sigma;
        float       sl = KIND == kGather ? s.gather_luma_sigma : s.eaw_luma_sigma;
        if (KIND == kBlur) sl = sl * sqrtf(fmaxf(0.0f, cvar + kEpsPost));
        k.s_normal = KIND == kGather ? s.gather_normal_sigma : s.eaw_normal_sigma;
        k.s_depth = k.cd * sd, k.s_luma = sl;  // (the stride-1 a-trous pass: cd * (float)1 * sigma = cd * sigma)
        k.f_depth = FAST ? fast_neg_inv(k.s_depth) : 0.f, k.f_luma = FAST ? fast_neg_inv(sl) : 0.f;
        constexpr bool LUMA = KIND != kBlur || USE_VAR;
        v3             filtered = mk3(0.f, 0.f, 0.f);
        float          total = 0.0f, a0 = 0.0f, a1 = 0.0f;  // a0: variance sum (Blur) or first moment (Disocclusion); a1: second moment
        const int      lc = ((int)(threadIdx.x >> 5) + R) * TW + (int)(threadIdx.x & 31u) + R;
        // exact mode: the divisions of the three weights in their unscaled form where the tile's texels and this wave's sigmas allow it
        // (div_unscaled); the IEEE form otherwise -- the same bits either way
        const bool wave_ok = !FAST && tile_ok && __ballot(!(div_sigma_ok(k.s_depth) && k.s_depth != 0.0f && div_sigma_ok(k.s_luma) && k.s_luma != 0.0f && div_depth_ok(k.cd) && k.cd >= 1e-5f)) == 0ull;
        auto taps = [&](auto ns_tag) {
            constexpr bool NS = decltype(ns_tag)::value;
        // fast mode: fully unrolled (offsets, tap lengths and kernel weights fold into immediates; ~25 instructions per tap).  Exact
        // mode: one row of taps per iteration -- at ~150 instructions per tap a fully unrolled 7 x 7 body would be 50 KB of code that
        // every wave streams through once, against a 64-KB instruction cache shared by two CUs
        constexpr int UNROLL_Y = FAST ? 2 * R + 1 : 1;
#pragma unroll UNROLL_Y
        for (int dy = -R; dy <= R; ++dy)
#pragma unroll
            for (int dx = -R; dx <= R; ++dx)
            {
                const int    e = lc + dy * TW + dx;
                const float4 v = t_col[e], g = t_nd[e];
                const v3     c = xyz(v);
                float        wgt, lw;
                tap_weights<FAST, LUMA, NS>(k, g, c, kLen7.v[dy + 3][dx + 3], kInv7.v[dy + 3][dx + 3], wgt, lw, dx == 0 && dy == 0);
                wgt = (g.w < 1e-5f) ? 0.0f : wgt;
                if (KIND == kBlur)
                {
                    const float hw = USE_VAR ? atrous_kernel(dx) * atrous_kernel(dy) : 1.0f;
                    const float kk = FAST ? wgt * hw : wgt * hw * lw;
                    filtered = acc3<FAST>(filtered, c, kk);
                    total += kk;
                    if (USE_VAR) a0 = FAST ? fmaf(kk * kk, v.w, a0) : a0 + hw * hw * wgt * wgt * lw * lw * v.w;
                }
                else
                {
                    const float w = FAST ? wgt : wgt * lw;
                    filtered = acc3<FAST>(filtered, c, w);
                    total += w;
                    if (KIND == kDisocclusion)
                    {
                        const float2 m = t_mom[e];
                        if (FAST)
                            a0 = fmaf(w, m.x, a0), a1 = fmaf(w, m.y, a1);
                        else
                            a0 += w * m.x, a1 += w * m.y;
                    }
                }
            }
        };
        if (wave_ok)
            taps(std::true_type{});
        else
            taps(std::false_type{});
        const bool  empty = total < kEpsPost;
        const float rt    = FAST ? fast_rcp(total) : 0.f;
        const v3    r     = empty ? k.cc : (FAST ? filtered * rt : div3(filtered, total));
        float       rw    = res.w;
        if (KIND == kBlur) rw = empty ? cvar : (FAST ? a0 * (rt * rt) : a0 / (total * total));
        if (KIND == kDisocclusion)
        {
            const float m0 = empty ? 0.0f : (FAST ? a0 * rt : a0 / total), m1 = empty ? 0.0f : (FAST ? a1 * rt : a1 / total);
            const float boost = FAST ? 8.0f * fast_rcp(hist) : 8.0f / hist;
            rw                = boost * fabsf(m1 - m0 * m0);
        }
        res = make_float4(r.x, r.y, r.z, rw);
    }
    out[o] = res;
}

// The a-trous passes of stride 3, 5, 7: 64 columns x 8 rows of one row phase per 512-thread workgroup (see the block comment above).
// COMBINE: the chain's last a-trous pass also does CombineIllumination (combine_illumination.hlsl:16-30, type 0) on its result --
// the operations of k_combine on the same operands, so the same bits, and the blurred image is neither written nor read back.
constexpr int kPhaseW = 64, kPhaseH = 8;
inline uint32_t phase_tiles_per_phase(uint32_t h, uint32_t stride) { return ((h + stride - 1) / stride + kPhaseH - 1) / kPhaseH; }
template <int STRIDE, bool USE_VAR, bool FAST, bool COMBINE>
__global__ __launch_bounds__(kPhaseW * kPhaseH) void k_blur_phase(PostSettingsDev s, Img color, Img nd, float4* out, const float4* albedo,
                                                                    const float4* direct, uint32_t aux_tiles_x, uint32_t tiles_x, uint32_t tiles_per_phase)
{
    constexpr int HX = 2 * STRIDE, TW = kPhaseW + 2 * HX, TH = kPhaseH + 4;
    __shared__ float4 t_col[TW * TH], t_nd[TW * TH];
    const int W = (int)color.w, H = (int)color.h;
    uint32_t  t;
    if (!xcd_tile_index(tiles_x * tiles_per_phase * (uint32_t)STRIDE, t)) return;  // workgroup-uniform
    // linear tile index: columns fastest, then the 8-row groups of a phase (neighbours share 4 of their 12 staged rows), then the phase
    const uint32_t tx = t % tiles_x, rest = t / tiles_x, jt = rest % tiles_per_phase, py = rest / tiles_per_phase;
    const int      x0 = (int)tx * kPhaseW - HX, j0 = (int)jt * kPhaseH - 2;
    bool           bad = false;
    for (int e = (int)threadIdx.x; e < TW * TH; e += kPhaseW * kPhaseH)
    {
        const int ty = e / TW, txx = e - ty * TW;
        float4    c, g;
        float2    mm;
        bool      ok = true;
#if defined(CAP_POST_DIAG) && CAP_POST_DIAG == 1  // diagnostic: no staging loads
        c = make_float4(0.5f, 0.4f, 0.3f, 0.1f), g = make_float4(0.f, 0.f, 1.f, 1.f + 0.001f * (float)txx);
#else
        stage_texel<true, false>(color, nd, nd, x0 + txx, (int)py + STRIDE * (j0 + ty), c, g, mm, 0, nullptr, FAST ? nullptr : &ok);
#endif
        bad |= !ok;
        t_col[e] = c, t_nd[e] = g;
    }
    bool tile_ok = true;  // (see k_stencil_lds)
    if (FAST)
        __syncthreads();
    else
        tile_ok = __syncthreads_or(bad ? 1 : 0) == 0;
    const int lx = (int)(threadIdx.x % kPhaseW), ly = (int)(threadIdx.x / kPhaseW);
    const int x = (int)tx * kPhaseW + lx, y = (int)py + STRIDE * ((int)jt * kPhaseH + ly);
    if (x >= W || y >= H) return;
    const size_t o  = (size_t)y * W + x;
    const float4 cv = color.p[o], cg = nd.p[o];  // (a background centre passes its own colour through: not from the tile)
    TapCenter    k;
    k.cn = xyz(cg), k.cd = cg.w, k.cc = remove_fireflies(cv), k.lcc = luminance(k.cc);
    const float cvar = USE_VAR ? cv.w : 0.0f;
    float4      res  = make_float4(k.cc.x, k.cc.y, k.cc.z, cvar);
#if defined(CAP_POST_DIAG) && CAP_POST_DIAG == 2  // diagnostic: no taps
    if (k.cd < -1.0f)
#else
    if (!(k.cd < 1e-5f))
#endif
    {
        k.s_normal = s.eaw_normal_sigma;
        k.s_depth  = k.cd * (float)STRIDE * s.eaw_depth_sigma;
        k.s_luma   = s.eaw_luma_sigma * sqrtf(fmaxf(0.0f, cvar + kEpsPost));
        k.f_depth = FAST ? fast_neg_inv(k.s_depth) : 0.f, k.f_luma = FAST ? fast_neg_inv(k.s_luma) : 0.f;
        v3          filtered = mk3(0.f, 0.f, 0.f);
        float       total = 0.0f, fvar = 0.0f;
        const int   lc = (ly + 2) * TW + lx + HX;
        const bool wave_ok = !FAST && tile_ok && __ballot(!(div_sigma_ok(k.s_depth) && k.s_depth != 0.0f && div_sigma_ok(k.s_luma) && k.s_luma != 0.0f && div_depth_ok(k.cd))) == 0ull;
        auto taps = [&](auto ns_tag) {
            constexpr bool NS = decltype(ns_tag)::value;
        constexpr int UNROLL_Y = FAST ? 5 : 1;  // (see k_stencil_lds)
#pragma unroll UNROLL_Y
        for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx)
            {
                const int    e = lc + dy * TW + dx * STRIDE;
                const float4 v = t_col[e], g = t_nd[e];
                const v3     c = xyz(v);
                float        wgt, lw;
                tap_weights<FAST, USE_VAR, NS>(k, g, c, kLen7.v[dy + 3][dx + 3], kInv7.v[dy + 3][dx + 3], wgt, lw, dx == 0 && dy == 0);
                wgt = (g.w < 1e-5f) ? 0.0f : wgt;
                const float hw = USE_VAR ? atrous_kernel(dx) * atrous_kernel(dy) : 1.0f;
                const float kk = FAST ? wgt * hw : wgt * hw * lw;
                filtered = acc3<FAST>(filtered, c, kk);
                total += kk;
                if (USE_VAR) fvar = FAST ? fmaf(kk * kk, v.w, fvar) : fvar + hw * hw * wgt * wgt * lw * lw * v.w;
            }
        };
        if (wave_ok)
            taps(std::true_type{});
        else
            taps(std::false_type{});
        const bool  empty = total < kEpsPost;
        const float rt    = FAST ? fast_rcp(total) : 0.f;
        const v3    r     = empty ? k.cc : (FAST ? filtered * rt : div3(filtered, total));
        res = make_float4(r.x, r.y, r.z, empty ? cvar : (FAST ? fvar * (rt * rt) : fvar / (total * total)));
    }
    if (COMBINE)
    {
        const size_t oa = aux_index((uint32_t)x, (uint32_t)y, color.w, aux_tiles_x);
        const float4 a = albedo[oa], d = direct[oa];
        res = make_float4(res.x * a.x + d.x, res.y * a.y + d.y, res.z * a.z + d.z, 1.0f * a.w + d.w);
    }
    out[o] = res;
}

// combine_illumination.hlsl:16-40, in place.  type = SettingsComponent::output (raytracing_system.cpp:1415): 0 combined, 1 direct,
// 2 indirect (w = 1), 3 the indirect image's variance channel
__global__ __launch_bounds__(kBlock) void k_combine(float4* io, const float4* albedo, const float4* direct, uint32_t n, uint32_t w, uint32_t aux_tiles_x,
                                                    int type)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    {
        const size_t oa = aux_tiles_x ? aux_index(i % w, i / w, w, aux_tiles_x) : (size_t)i;
        const float4 in = io[i];
        if (type == 0)
        {
            const float4 a = albedo[oa], d = direct[oa];
            io[i] = make_float4(in.x * a.x + d.x, in.y * a.y + d.y, in.z * a.z + d.z, 1.0f * a.w + d.w);
        }
        else if (type == 1)
            io[i] = direct[oa];
        else if (type == 2)
            io[i] = make_float4(in.x, in.y, in.z, 1.0f);
        else
            io[i] = make_float4(in.w, in.w, in.w, 1.0f);
    }
}

// aabb.h:24-34
__device__ __forceinline__ v3 clip_to_aabb(v3 pmin, v3 pmax, v3 p)
{
    const v3    c = (pmin + pmax) * 0.5f, radius = (pmax - pmin) * 0.5f, dc = p - c;
    const v3    clip = mk3(dc.x / (radius.x + 1e-5f), dc.y / (radius.y + 1e-5f), dc.z / (radius.z + 1e-5f));
    const float m    = fmaxf(fmaxf(fabsf(clip.x), fabsf(clip.y)), fabsf(clip.z));
    return m > 1.0f ? c + div3(dc, m) : p;
}

// temporal_accumulation.hlsl:362-447
// FAST (CapPostSettings::fast_weights): as in k_accumulate, what DECIDES (reprojection, velocity, the plain path) is the exact code;
// the values use v_rcp_f32 / v_sqrt_f32, the bilinear tap at a whole-pixel position is the mean of its 2 x 2 texels (its fractional
// weights are 0.5 up to one rounding error), the current colour at the pixel centre is the texel, and the history resample is the
// bicubic's centre tap.
constexpr uint32_t kTaaTileW = 32 + 4, kTaaTileH = 8 + 4;  // the workgroup's 32 x 8 pixels + the 5 x 5 window's halo
__device__ __forceinline__ v3 simple_tonemap_fast(v3 v) { return v * fast_rcp(1.0f + luminance(v)); }
__device__ __forceinline__ v3 rgb2ycocg_fast(v3 c) { return mk3(0.25f * c.x + 0.5f * c.y + 0.25f * c.z, 0.5f * c.x - 0.5f * c.z, -0.25f * c.x + 0.5f * c.y - 0.25f * c.z); }
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_taa(PostSettingsDev s, CameraDev cam, CameraDev prev_cam, Img color, Img nd, Img history_img,
                                                float4* out)
{
    const uint32_t W = color.w, H = color.h;
    __shared__ v3  lds_tap[kTaaTileW * kTaaTileH];
    // FAST: the pixel's own G-buffer texel and colour are requested before the tile is staged, the history's footprint right behind
    // the reprojection, and the plain path is a select at the end: one batch of loads per dependency level instead of one per branch
    uint32_t   x = blockIdx.x * 32u + (threadIdx.x & 31u), y = blockIdx.y * 8u + (threadIdx.x >> 5);
    const bool in_image = x < W && y < H;
    float4     g_early = make_float4(0.f, 0.f, 0.f, 0.f);
    v3         cur_early = mk3(0.f, 0.f, 0.f);
    if (FAST && in_image)
    {
        g_early   = nd.p[(size_t)y * W + x];
        cur_early = xyz(color.p[(size_t)y * W + x]);
    }
    {
        const int x0 = (int)(blockIdx.x * 32u) - 2, y0 = (int)(blockIdx.y * 8u) - 2;
        for (uint32_t e = threadIdx.x; e < kTaaTileW * kTaaTileH; e += kBlock)
        {
            int sx = x0 + (int)(e % kTaaTileW), sy = y0 + (int)(e / kTaaTileW);
            sx = sx < 0 ? 0 : (sx > (int)W - 1 ? (int)W - 1 : sx);
            sy = sy < 0 ? 0 : (sy > (int)H - 1 ? (int)H - 1 : sy);
            if (FAST)
            {
                // SampleBilinear at uv = (sx, sy) / dim: footprint (sx - 1 .. sx, sy - 1 .. sy) with weights 0.5; at the left / top
                // edge uint(floor(-0.5)) saturates to 0, so the footprint is (0 .. 1) there (utils.h:24-27)
                const uint32_t ux = sx > 0 ? (uint32_t)sx - 1u : 0u, uy = sy > 0 ? (uint32_t)sy - 1u : 0u;
                const v3 v00 = xyz(ld(color, ux, uy)), v01 = xyz(ld(color, ux, uy + 1)), v10 = xyz(ld(color, ux + 1, uy)), v11 = xyz(ld(color, ux + 1, uy + 1));
                lds_tap[e] = rgb2ycocg_fast(simple_tonemap_fast(((v00 + v10) + (v01 + v11)) * 0.25f));
            }
            else
                lds_tap[e] = rgb2ycocg(simple_tonemap(sample_bilinear(color, xy_to_uv(f2{(float)sx, (float)sy}, W, H))));
        }
        __syncthreads();
    }
    if (!in_image) return;
    const uint32_t lx = threadIdx.x & 31u, ly = threadIdx.x >> 5;
    const v3       tap_of_thread = lds_tap[(ly + 2) * kTaaTileW + (lx + 2)];
    const f2     uv = f2{((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
    const float4 g  = FAST ? g_early : ld(nd, x, y);
    const size_t o  = (size_t)y * W + x;
    bool         plain = g.w < 1e-5f;
    f2           puv = f2{0.5f, 0.5f};
    float        velocity = 0.0f;
    if (FAST || !plain)
    {
        const v3 hit = reconstruct_world_position(cam, uv, (FAST && plain) ? 1.0f : g.w);
        const f2 p   = image_plane_uv(prev_cam, hit);
        const float vx = (p.x - uv.x) * (float)W, vy = (p.y - uv.y) * (float)H;
        velocity     = sqrtf(fmaf(vy, vy, vx * vx));
        plain        = plain || p.x < 0.0f || p.y < 0.0f || p.x > 1.0f || p.y > 1.0f;
        puv          = (FAST && plain) ? puv : p;  // (FAST: a position nothing is read for keeps an in-image dummy)
    }
    v3 history_rgb = mk3(0.f, 0.f, 0.f);
    if (FAST)
    {
        const f2 c0  = uv_to_xy(puv, W, H);
        const f2 cuv = f2{fminf(fmaxf(c0.x * fast_rcp((float)W), 0.0f), 1.0f), fminf(fmaxf(c0.y * fast_rcp((float)H), 0.0f), 1.0f)};
        history_rgb  = sample_bilinear_fast(history_img, cuv);
    }
    // (the last row and column are NOT the texel: UVtoXY clamps to dim - 1, utils.h:6-10, so the reference's tap there is a two-texel blend)
    const v3 cur = (FAST && x + 1u < W && y + 1u < H) ? cur_early : sample_bilinear(color, uv);
    if (!FAST && plain)
    {
        out[o] = make_float4(cur.x, cur.y, cur.z, 1.0f);
        return;
    }
    const bool  is_static = velocity < 1e-3f;
    float       alpha = is_static ? 0.98f : 0.6f;
    const float scale = is_static ? 5.0f : 0.75f;
    alpha             = fminf(s.taa_feedback, alpha);
    v3 history, c;
    if (FAST)
    {
        history = rgb2ycocg_fast(simple_tonemap_fast(history_rgb));
        c       = rgb2ycocg_fast(simple_tonemap_fast(cur));
    }
    else
    {
        history = rgb2ycocg(simple_tonemap(resample_bicubic(history_img, puv)));
        c       = rgb2ycocg(simple_tonemap(cur));
    }
    // CalculateNeighbourhoodColorAABB(gidx, dim, scale), :98-137.  The 25 tonemapped bilinear taps of a pixel sit at whole-pixel
    // positions (clamped to the image), so a tap's value depends on that position only and neighbouring pixels share 20 of their
    // 25: the workgroup evaluates each position of its 36 x 12 footprint once into LDS (the same operations on the same
    // operands as the per-pixel evaluation, so the same bits) and every pixel sums its 5 x 5 window in the reference's order.
    const v3 center = tap_of_thread;
    v3       m1 = mk3(0.f, 0.f, 0.f), m2 = mk3(0.f, 0.f, 0.f);
    for (int i = -2; i <= 2; ++i)
        for (int j = -2; j <= 2; ++j)
        {
            const v3 v = lds_tap[(ly + 2 + j) * kTaaTileW + (lx + 2 + i)];
            m1 = m1 + v;
            m2 = FAST ? mk3(fmaf(v.x, v.x, m2.x), fmaf(v.y, v.y, m2.y), fmaf(v.z, v.z, m2.z)) : m2 + v * v;
        }
    const float inv_n = 1.0f / 25.0f;
    m1 = m1 * inv_n, m2 = m2 * inv_n;
    const v3 var = m2 - m1 * m1;
    const v3 dev = mk3(sqrtf(fabsf(var.x)) * scale, sqrtf(fabsf(var.y)) * scale, sqrtf(fabsf(var.z)) * scale);
    const v3 lo = m1 - dev, hi = m1 + dev;
    const v3 pmin = mk3(fminf(lo.x, center.x), fminf(lo.y, center.y), fminf(lo.z, center.z));
    const v3 pmax = mk3(fmaxf(hi.x, center.x), fmaxf(hi.y, center.y), fmaxf(hi.z, center.z));
    if (FAST)
    {
        // aabb.h:24-34 and the inverse tonemap with v_rcp_f32
        const v3    cc = (pmin + pmax) * 0.5f, radius = (pmax - pmin) * 0.5f, dc = history - cc;
        const v3    clip = mk3(dc.x * fast_rcp(radius.x + 1e-5f), dc.y * fast_rcp(radius.y + 1e-5f), dc.z * fast_rcp(radius.z + 1e-5f));
        const float m    = fmaxf(fmaxf(fabsf(clip.x), fabsf(clip.y)), fabsf(clip.z));
        history          = m > 1.0f ? cc + dc * fast_rcp(m) : history;
        const v3 rr      = ycocg2rgb(lerp3(c, history, alpha));
        const v3 r       = rr * fast_rcp(1.0f - luminance(rr));
        out[o]           = plain ? make_float4(cur.x, cur.y, cur.z, 1.0f) : make_float4(r.x, r.y, r.z, 1.0f);
    }
    else
    {
        history    = clip_to_aabb(pmin, pmax, history);
        const v3 r = invert_simple_tonemap(ycocg2rgb(lerp3(c, history, alpha)));
        out[o]     = make_float4(r.x, r.y, r.z, 1.0f);
    }
}
// cap_debug_get(CAP_DEBUG_SELFTEST_DIV): div_unscaled against the compiler's IEEE division, bit for bit, on the device.
//   out[0]: log2_c_ns(x) != log2_c(x) over EVERY positive normal float x (the only division inside is (m - 1) / (m + 1));
//   out[1]: div_unscaled(a, b) != a / b over 2^30 pseudo-random pairs that cover the range it is used on: a = 0 or in [2^-80, 2^42),
//           b in [2^-40, 2^41) (what the call sites can pass: a sigma <= 2^38 times a tap length <= sqrt(18)), exponents and mantissas drawn independently.
__device__ __forceinline__ uint32_t selftest_hash(uint32_t x)
{
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(kBlock) void k_div_selftest(unsigned long long* out)
{
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x, total = gridDim.x * kBlock;
    unsigned long long bad0 = 0, bad1 = 0;
    for (uint64_t b = 0x00800000ull + tid; b <= 0x7f7fffffull; b += total)
    {
        const float x = u2f((uint32_t)b);
        bad0 += f2u(log2_c_ns(x)) != f2u(log2_c(x)) ? 1u : 0u;
    }
    for (uint32_t i = tid; i < (1u << 30); i += total)
    {
        const uint32_t h0 = selftest_hash(i), h1 = selftest_hash(i ^ 0x9e3779b9u), h2 = selftest_hash(h0 + h1);
        const uint32_t ea = 47u + h2 % 122u, eb = 87u + (h2 >> 8) % 81u;  // exponent fields: 2^-80 .. 2^41, 2^-40 .. 2^40
        const float    a = (h2 >> 26) == 0u ? 0.0f : u2f((ea << 23) | (h0 & 0x007fffffu)), bb = u2f((eb << 23) | (h1 & 0x007fffffu));
        bad1 += f2u(div_unscaled(a, bb)) != f2u(a / bb) ? 1u : 0u;
    }
    if (bad0) atomicAdd(&out[0], bad0);
    if (bad1) atomicAdd(&out[1], bad1);
    // positive control: how many comparisons this thread made -- the host fails the self-test unless the total is the
    // number of normal floats + 2^30 (a kernel that never ran leaves zeros everywhere and used to read as "passed")
    unsigned long long n = 0;
    for (uint64_t b = 0x00800000ull + tid; b <= 0x7f7fffffull; b += total) ++n;
    for (uint32_t i = tid; i < (1u << 30); i += total) ++n;
    atomicAdd(&out[2], n);
}
}  // namespace

void launch_div_selftest(hipStream_t stream, unsigned long long* out_device)
{
    hipLaunchKernelGGL(k_div_selftest, dim3(4096), dim3(kBlock), 0, stream, out_device);
}

void launch_post_chain(hipStream_t stream, const PostChainArgs& a)
{
    const uint32_t W = a.width, H = a.height;
    const dim3     grid((W + 31) / 32, (H + 7) / 8), block(kBlock);
    const dim3     xgrid(xcd_grid(((W + 31) / 32) * ((H + 7) / 8)));  // the 32 x 8 LDS-tiled kernels' 1-D grid (xcd_tile)
    const size_t   bytes = sizeof(float4) * (size_t)W * H;
    auto           img   = [&](const float4* p) { return Img{p, W, H}; };
    const Img      none{nullptr, 0, 0};
    const uint32_t src = (a.frame_count + 1) % 2, dst = a.frame_count % 2;  // raytracing_system.cpp:1709-1710, 1754-1755
    uint32_t cg = (W * H + kBlock - 1) / kBlock;  // grid of the streaming (stencil-free) kernels
    if (cg > 4096) cg = 4096;
    if (cg == 0) cg = 1;
    auto mark = [&](int pass) {
        if (a.mark) a.mark(a.mark_user, pass);
    };
    mark(0);
    // decoded (normal.xyz, depth) image of the frame: every later pass reads the G-buffer through it (the depth in .w is the raw
    // texel's), and it is what the next frame keeps as its previous normal/depth image
    const float4* indirect = a.indirect;
    const bool     up = a.settings.lowres_indirect != 0;
    const bool     fast = a.settings.fast_weights != 0, use_var = a.settings.use_variance != 0;
    // Gather straight from the render's tile-ordered planes (full-resolution indirect pass): untiling and normal decoding happen in
    // its staging, the decoded image is its second output
    const bool gather_tiled = a.tiled && a.tiled_indirect && a.settings.gather && !up;
    if (gather_tiled)
        ;
    else if (a.tiled)
    {
        hipLaunchKernelGGL(k_untile_decode, dim3(cg), block, 0, stream, a.screen, a.tiled_indirect, a.tiled_normal_depth, a.indirect_rowmajor, a.normals);
        if (a.tiled_indirect) indirect = a.indirect_rowmajor;
    }
    else
        hipLaunchKernelGGL(k_decode_normals, dim3(cg), block, 0, stream, a.normal_depth, a.normals, W * H);
    // SpatialGather (cpp:1541-1604); with lowres_indirect the input, the grid and indirect_temp are (W/2, H/2)
    const uint32_t IW = up ? W >> 1 : W, IH = up ? H >> 1 : H;
    const Img      indirect_in{indirect, IW, IH};
    const int      ox = (int)((a.frame_count % 4u) / 2u), oy = (int)((a.frame_count % 4u) % 2u);
    if (gather_tiled)
    {
        const Img tc{a.tiled_indirect, W, H}, tn{a.tiled_normal_depth, W, H};
        if (fast)
            hipLaunchKernelGGL((k_stencil_lds<kGather, 3, true, true, true>), xgrid, block, 0, stream, a.settings, tc, tn, none, a.indirect_temp, a.tiled, a.normals);
        else
            hipLaunchKernelGGL((k_stencil_lds<kGather, 3, true, false, true>), xgrid, block, 0, stream, a.settings, tc, tn, none, a.indirect_temp, a.tiled, a.normals);
    }
    else if (a.settings.gather && up)
    {
        if (fast)
            hipLaunchKernelGGL((k_gather<true, true>), dim3((IW + 31) / 32, (IH + 7) / 8), block, 0, stream, a.settings, indirect_in, img(a.normals),
                               a.indirect_temp, ox, oy);
        else
            hipLaunchKernelGGL((k_gather<true, false>), dim3((IW + 31) / 32, (IH + 7) / 8), block, 0, stream, a.settings, indirect_in, img(a.normals),
                               a.indirect_temp, ox, oy);
    }
    else if (a.settings.gather)
    {
        if (fast)
            hipLaunchKernelGGL((k_stencil_lds<kGather, 3, true, true>), xgrid, block, 0, stream, a.settings, indirect_in, img(a.normals), none, a.indirect_temp);
        else
            hipLaunchKernelGGL((k_stencil_lds<kGather, 3, true, false>), xgrid, block, 0, stream, a.settings, indirect_in, img(a.normals), none, a.indirect_temp);
    }
    else
        (void)hipMemcpyAsync(a.indirect_temp, indirect, sizeof(float4) * (size_t)IW * IH, hipMemcpyDeviceToDevice, stream);
    // IntegrateTemporally (cpp:1283-1342)
    mark(1);
    if (fast)
        hipLaunchKernelGGL(k_accumulate<true>, grid, block, 0, stream, a.settings, a.frame_count, a.camera, a.prev_camera,
                           Img{a.indirect_temp, IW, IH}, img(a.normals), img(a.indirect_history[src]), img(a.moments_history[src]), img(a.prev_normal_depth),
                           a.indirect_history[dst], a.moments_history[dst]);
    else
        hipLaunchKernelGGL(k_accumulate<false>, grid, block, 0, stream, a.settings, a.frame_count, a.camera, a.prev_camera,
                           Img{a.indirect_temp, IW, IH}, img(a.normals), img(a.indirect_history[src]), img(a.moments_history[src]), img(a.prev_normal_depth),
                           a.indirect_history[dst], a.moments_history[dst]);
    // Denoise (cpp:1437-1538)
    mark(2);
    if (a.settings.denoise)
    {
        // (USE_VAR, FAST) variants; the last a-trous pass combines (its own timestamp label then covers nothing: the reference's
        // "Combine illumination" span is part of "EAW" here)
        const bool fuse_combine = a.settings.output == 0;  // the other output types read the blurred image itself: k_combine after the pass
#define CAP_MODES(KERNEL_UV_F, ...)                                                  \
    {                                                                                 \
        if (use_var && fast) { KERNEL_UV_F(true, true, __VA_ARGS__); }                \
        else if (use_var) { KERNEL_UV_F(true, false, __VA_ARGS__); }                  \
        else if (fast) { KERNEL_UV_F(false, true, __VA_ARGS__); }                     \
        else { KERNEL_UV_F(false, false, __VA_ARGS__); }                              \
    }
#define CAP_STENCIL(UV, F, KIND, R, IN, MOM, OUT) hipLaunchKernelGGL((k_stencil_lds<KIND, R, UV, F>), xgrid, block, 0, stream, a.settings, IN, img(a.normals), MOM, OUT)
#define CAP_STENCIL_DUAL(UV, F, IN, MOM, OUT, ALT) \
    hipLaunchKernelGGL((k_stencil_lds<kBlur, 2, UV, F, false, true>), xgrid, block, 0, stream, a.settings, IN, img(a.normals), MOM, OUT, 0u, nullptr, ALT)
#define CAP_PHASE(UV, F, S, CB, IN, OUT)                                                                                                        \
    hipLaunchKernelGGL((k_blur_phase<S, UV, F, CB>), pgrid, pblock, 0, stream, a.settings, IN, img(a.normals), OUT, (CB) ? a.albedo : nullptr,   \
                       (CB) ? a.direct : nullptr, a.tiled, ptx, ptp)
        auto blur = [&](uint32_t stride, const float4* in, float4* out, bool last_pass) {
            const bool last = last_pass && fuse_combine;
            if (stride == 1u)
            {
                // (`in` = BlurDisocclusion's sparse output: see stage_texel DUAL)
                CAP_MODES(CAP_STENCIL_DUAL, img(a.indirect_history[dst]), img(a.moments_history[dst]), out, in);
                return;
            }
            const uint32_t ptx = (W + kPhaseW - 1) / kPhaseW, ptp = phase_tiles_per_phase(H, stride);
            const dim3     pgrid(xcd_grid(ptx * ptp * stride)), pblock(kPhaseW * kPhaseH);
#define CAP_PHASE_S(S)                                                     \
    {                                                                      \
        if (last) CAP_MODES(CAP_PHASE, S, true, img(in), out)              \
        else CAP_MODES(CAP_PHASE, S, false, img(in), out)                  \
    }
            if (stride == 3u) CAP_PHASE_S(3)
            else if (stride == 5u) CAP_PHASE_S(5)
            else CAP_PHASE_S(7)
#undef CAP_PHASE_S
        };
        CAP_MODES(CAP_STENCIL, kDisocclusion, 3, img(a.indirect_history[dst]), img(a.moments_history[dst]), a.temp[0]);
        blur(1u, a.temp[0], a.temp[1], false);
        blur(3u, a.temp[1], a.temp[0], !a.settings.eaw5);
        if (a.settings.eaw5)
        {
            blur(5u, a.temp[0], a.temp[1], false);
            blur(7u, a.temp[1], a.temp[0], true);
        }
#undef CAP_PHASE
#undef CAP_STENCIL_DUAL
#undef CAP_STENCIL
#undef CAP_MODES
        mark(3);
        if (!fuse_combine) hipLaunchKernelGGL(k_combine, dim3(cg), block, 0, stream, a.temp[0], a.albedo, a.direct, W * H, W, a.tiled, a.settings.output);
    }
    else
    {
        (void)hipMemcpyAsync(a.temp[0], a.indirect_history[dst], bytes, hipMemcpyDeviceToDevice, stream);
        // CombineIllumination (cpp:1400-1435)
        mark(3);
        hipLaunchKernelGGL(k_combine, dim3(cg), block, 0, stream, a.temp[0], a.albedo, a.direct, W * H, W, a.tiled, a.settings.output);
    }
    // ApplyTAA (cpp:1344-1398)
    mark(4);
    if (a.settings.fast_weights)
        hipLaunchKernelGGL(k_taa<true>, grid, block, 0, stream, a.settings, a.camera, a.prev_camera, img(a.temp[0]), img(a.normals),
                           img(a.combined_history[src]), a.combined_history[dst]);
    else
        hipLaunchKernelGGL(k_taa<false>, grid, block, 0, stream, a.settings, a.camera, a.prev_camera, img(a.temp[0]), img(a.normals),
                           img(a.combined_history[src]), a.combined_history[dst]);
    // CopyGBuffer of the next frame (cpp:955-1009): the caller makes `normals` the next call's `prev_normal_depth` (two buffers
    // changing roles; copying the image cost 7 us per 1080p frame)
    mark(5);
}

void launch_decimate2x(hipStream_t stream, const float4* full, uint32_t width, uint32_t height, uint32_t ox, uint32_t oy, float4* out)
{
    const uint32_t n = (width >> 1) * (height >> 1);
    uint32_t       g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (g == 0) g = 1;
    hipLaunchKernelGGL(k_decimate2x, dim3(g), dim3(kBlock), 0, stream, full, width, height, ox, oy, out);
}
}  // namespace cap
