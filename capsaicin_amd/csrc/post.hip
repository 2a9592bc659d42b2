// post.hip — gfx950 kernels of the reference's reconstruction chain, the stage right after the ray passes:
//   Gather -> Accumulate -> BlurDisocclusion -> Blur x2|x4 -> Combine -> TAA
// Reference (paths relative to /root/reference/src/core): shaders/spatial_gather.hlsl, temporal_accumulation.hlsl, eaw_blur.hlsl,
// combine_illumination.hlsl, eaw_edge_stopping.h, aabb.h, color_space.h, math_functions.h, utils.h, camera.h; pass order and
// buffer wiring src/systems/raytracing_system.cpp:262-317, 1283-1604, 1700-1790.  Full-resolution configuration
// (UPSCALE2X off, CALCULATE_VARIANCE / USE_VARIANCE on).  Buffers are fp32 float4 row-major images (the reference stores RGBA16F).
// Pure stencil / streaming work: HBM- and L2-bound, no MFMA.
#include "cap_kernels.h"
#include "cap_reproject.h"

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

// exp(x), x <= 0, and pow(x, s), x in [0,1]: the exp2/log2 polynomials of the arithmetic contract (cap_math.h)
__device__ __forceinline__ float exp_neg(float x)
{
    const float y = x * 1.44269504088896341f;
    if (!(y >= -125.0f)) return 0.0f;
    return exp2_c(y);
}
__device__ __forceinline__ float pow01(float x, float s)
{
    if (!(x >= 1.17549435e-38f)) return 0.0f;
    const float y = s * log2_c(x);
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
__device__ __forceinline__ v3 resample_bicubic(const Img& t, f2 uv)
{
    v3       filtered = mk3(0.f, 0.f, 0.f);
    const f2 c        = uv_to_xy(uv, t.w, t.h);
    float    tw       = 0.0f;
    for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j)
        {
            const f2 cur = f2{c.x + (float)i, c.y + (float)j};
            if (cur.x < 0.0f || cur.y < 0.0f || cur.x >= (float)t.w || cur.y >= (float)t.h) continue;
            const float kxy = cubic(fabsf(cur.x - c.x), 0.0f, 0.5f) * cubic(fabsf(cur.y - c.y), 0.0f, 0.5f);
            if (kxy == 0.0f) continue;
            const v3    value = sample_bilinear(t, xy_to_uv(cur, t.w, t.h));
            const float w     = kxy * (1.0f / (1.0f + luminance(value)));
            filtered = filtered + value * w;
            tw += w;
        }
    return tw > 1e-5f ? div3(filtered, tw) : mk3(0.f, 0.f, 0.f);
}
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
// eaw_edge_stopping.h
__device__ __forceinline__ float normal_weight(v3 n0, v3 n1, float s) { return pow01(fmaxf(dot3(n0, n1), 0.0f), s); }
__device__ __forceinline__ float depth_weight(float dc, float dp, float s)
{
    const float t = s == 0.0f ? 0.0f : (fabsf(dc - dp) / s);
    return exp_neg(-t);
}
__device__ __forceinline__ float luma_weight(float lc, float lp, float s) { return exp_neg(-(fabsf(lc - lp) / s)); }
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

// spatial_gather.hlsl:28-109.  nd = decoded (normal.xyz, depth) image of k_decode_normals.
// UP (UPSCALE2X, :36-46, :83-87): the grid and `color` are half resolution and the G-buffer is read at (xy << 1) + (ox, oy).
// The taps are bounded by the FULL window size, as the host passes it (raytracing_system.cpp:1562-1569): a tap beyond the
// half-resolution image reads a G-buffer texel outside the window, i.e. depth 0, and is skipped as background.
template <bool UP>
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
                const float len = kLen7.v[dy + 3][dx + 3];
                const float wgt = normal_weight(cn, n, s_normal) * depth_weight(cd, g.w, s_depth * len) * luma_weight(luminance(cc), luminance(c), s_luma);
                filtered = filtered + c * wgt;
                total += wgt;
            }
        const v3 r = (total < kEpsPost) ? cc : div3(filtered, total);
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

// temporal_accumulation.hlsl:213-325.  With UPSCALE2X (s.lowres_indirect) `color` is the half-resolution image (SampleColor uses its
// size, :228-235) and a pixel that got no new sample this frame keeps its history (:307-313).
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
    float          alpha    = s.temporal_upscale_feedback;
    const v3       history  = resample_bicubic(color_history, puv);
    uint32_t       hist_len = sat_uint(ld(moments_history, sat_uint(floorf(pxy.x)), sat_uint(floorf(pxy.y))).w);
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
    const v3    mh = resample_bicubic(moments_history, puv);
    const float m0 = lerp1(l, mh.x, alpha), m1 = lerp1(l * l, mh.y, alpha);
    const float variance = fabsf(m1 - m0 * m0);
    out_moments[o]   = make_float4(m0, m1, 0.0f, (float)(hist_len + 1));
    const v3 blended = lerp3(c, history, alpha);
    out_color[o]     = make_float4(blended.x, blended.y, blended.z, variance);
}

__device__ __forceinline__ v3 remove_fireflies(float4 v) { return mk3(fminf(v.x, 10.0f), fminf(v.y, 10.0f), fminf(v.z, 10.0f)); }

// eaw_blur.hlsl:142-223
__global__ __launch_bounds__(kBlock) void k_blur_disocclusion(PostSettingsDev s, Img color, Img nd, Img moments, float4* out)
{
    uint32_t x, y;
    if (!pixel_of_thread(color.w, color.h, x, y)) return;
    const float  hist = ld(moments, x, y).w;
    const float4 cg   = ld(nd, x, y);
    const v3     cn   = xyz(cg);
    const float  cd   = cg.w;
    const float4 cv   = ld(color, x, y);
    const v3     cc   = remove_fireflies(cv);
    float4       res  = make_float4(cc.x, cc.y, cc.z, cv.w);
    if (!(cd < 1e-5f || hist >= 8.0f))
    {
        const float s_depth = cd * s.eaw_depth_sigma, s_normal = s.eaw_normal_sigma, s_luma = s.eaw_luma_sigma;
        v3          filtered = mk3(0.f, 0.f, 0.f);
        float       fm0 = 0.0f, fm1 = 0.0f, total = 0.0f;
        for (int dy = -3; dy <= 3; ++dy)
            for (int dx = -3; dx <= 3; ++dx)
            {
                const int sx = (int)x + dx, sy = (int)y + dy;
                if (sx < 0 || sy < 0 || sx >= (int)color.w || sy >= (int)color.h) continue;
                const v3     c = remove_fireflies(ldi(color, sx, sy));
                const float4 g = ldi(nd, sx, sy);
                const float4 m = ldi(moments, sx, sy);
                if (g.w < 1e-5f) continue;
                const v3    n   = xyz(g);
                const float len = kLen7.v[dy + 3][dx + 3];
                const float wgt = normal_weight(cn, n, s_normal) * depth_weight(cd, g.w, s_depth * len) * luma_weight(luminance(cc), luminance(c), s_luma);
                filtered = filtered + c * wgt;
                fm0 += wgt * m.x, fm1 += wgt * m.y;
                total += wgt;
            }
        const v3    r  = (total < kEpsPost) ? cc : div3(filtered, total);
        const float m0 = (total < kEpsPost) ? 0.0f : fm0 / total, m1 = (total < kEpsPost) ? 0.0f : fm1 / total;
        const float boost = 8.0f / hist;
        res = make_float4(r.x, r.y, r.z, boost * fabsf(m1 - m0 * m0));
    }
    out[(size_t)y * color.w + x] = res;
}

// eaw_blur.hlsl:48-137
__global__ __launch_bounds__(kBlock) void k_blur(PostSettingsDev s, uint32_t stride, Img color, Img nd, float4* out)
{
    uint32_t x, y;
    if (!pixel_of_thread(color.w, color.h, x, y)) return;
    const float4 cg   = ld(nd, x, y);
    const v3     cn   = xyz(cg);
    const float  cd   = cg.w;
    const float4 cv   = ld(color, x, y);
    const v3     cc   = remove_fireflies(cv);
    const float  cvar = cv.w;
    float4       res  = make_float4(cc.x, cc.y, cc.z, cvar);
    if (!(cd < 1e-5f))
    {
        const float kw[3]   = {1.0f, 2.0f / 3.0f, 1.0f / 6.0f};
        const float s_depth = cd * (float)stride * s.eaw_depth_sigma, s_normal = s.eaw_normal_sigma;
        const float s_luma  = s.eaw_luma_sigma * sqrtf(fmaxf(0.0f, cvar + kEpsPost));
        v3          filtered = mk3(0.f, 0.f, 0.f);
        float       fvar = 0.0f, total = 0.0f;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx)
            {
                const int sx = (int)x + dx * (int)stride, sy = (int)y + dy * (int)stride;
                if (sx < 0 || sy < 0 || sx >= (int)color.w || sy >= (int)color.h) continue;
                const float4 v = ldi(color, sx, sy);
                const v3     c = remove_fireflies(v);
                const float4 g = ldi(nd, sx, sy);
                if (g.w < 1e-5f) continue;
                const v3    n   = xyz(g);
                const float lw  = luma_weight(luminance(cc), luminance(c), s_luma);
                const float hw  = kw[dx < 0 ? -dx : dx] * kw[dy < 0 ? -dy : dy];
                const float len = sqrtf((float)(dx * dx + dy * dy));
                const float wgt = normal_weight(cn, n, s_normal) * depth_weight(cd, g.w, s_depth * len);
                const float k   = wgt * hw * lw;
                filtered = filtered + c * k;
                total += k;
                fvar += hw * hw * wgt * wgt * lw * lw * v.w;
            }
        const v3    r  = (total < kEpsPost) ? cc : div3(filtered, total);
        const float rv = (total < kEpsPost) ? cvar : fvar / (total * total);
        res = make_float4(r.x, r.y, r.z, rv);
    }
    out[(size_t)y * color.w + x] = res;
}

// combine_illumination.hlsl:16-30, type 0, in place
__global__ __launch_bounds__(kBlock) void k_combine(float4* io, const float4* albedo, const float4* direct, uint32_t n)
{
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    {
        const float4 in = io[i], a = albedo[i], d = direct[i];
        io[i] = make_float4(in.x * a.x + d.x, in.y * a.y + d.y, in.z * a.z + d.z, 1.0f * a.w + d.w);
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
constexpr uint32_t kTaaTileW = 32 + 4, kTaaTileH = 8 + 4;  // the workgroup's 32 x 8 pixels + the 5 x 5 window's halo
__global__ __launch_bounds__(kBlock) void k_taa(PostSettingsDev s, CameraDev cam, CameraDev prev_cam, Img color, Img nd, Img history_img,
                                                float4* out)
{
    const uint32_t W = color.w, H = color.h;
    __shared__ v3  lds_tap[kTaaTileW * kTaaTileH];
    {
        const int x0 = (int)(blockIdx.x * 32u) - 2, y0 = (int)(blockIdx.y * 8u) - 2;
        for (uint32_t e = threadIdx.x; e < kTaaTileW * kTaaTileH; e += kBlock)
        {
            int sx = x0 + (int)(e % kTaaTileW), sy = y0 + (int)(e / kTaaTileW);
            sx = sx < 0 ? 0 : (sx > (int)W - 1 ? (int)W - 1 : sx);
            sy = sy < 0 ? 0 : (sy > (int)H - 1 ? (int)H - 1 : sy);
            lds_tap[e] = rgb2ycocg(simple_tonemap(sample_bilinear(color, xy_to_uv(f2{(float)sx, (float)sy}, W, H))));
        }
        __syncthreads();
    }
    uint32_t x, y;
    if (!pixel_of_thread(W, H, x, y)) return;
    const uint32_t lx = threadIdx.x & 31u, ly = threadIdx.x >> 5;
    const v3       tap_of_thread = lds_tap[(ly + 2) * kTaaTileW + (lx + 2)];
    const f2     uv = f2{((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
    const float4 g  = ld(nd, x, y);
    const size_t o  = (size_t)y * W + x;
    bool         plain = g.w < 1e-5f;
    f2           puv = f2{0.f, 0.f};
    float        velocity = 0.0f;
    if (!plain)
    {
        const v3 hit = reconstruct_world_position(cam, uv, g.w);
        puv          = image_plane_uv(prev_cam, hit);
        const float vx = (puv.x - uv.x) * (float)W, vy = (puv.y - uv.y) * (float)H;
        velocity     = sqrtf(fmaf(vy, vy, vx * vx));
        plain        = puv.x < 0.0f || puv.y < 0.0f || puv.x > 1.0f || puv.y > 1.0f;
    }
    const v3 cur = sample_bilinear(color, uv);
    if (plain)
    {
        out[o] = make_float4(cur.x, cur.y, cur.z, 1.0f);
        return;
    }
    const bool  is_static = velocity < 1e-3f;
    float       alpha = is_static ? 0.98f : 0.6f;
    const float scale = is_static ? 5.0f : 0.75f;
    alpha             = fminf(s.taa_feedback, alpha);
    v3       history = rgb2ycocg(simple_tonemap(resample_bicubic(history_img, puv)));
    const v3 c       = rgb2ycocg(simple_tonemap(cur));
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
            m2 = m2 + v * v;
        }
    const float inv_n = 1.0f / 25.0f;
    m1 = m1 * inv_n, m2 = m2 * inv_n;
    const v3 var = m2 - m1 * m1;
    const v3 dev = mk3(sqrtf(fabsf(var.x)) * scale, sqrtf(fabsf(var.y)) * scale, sqrtf(fabsf(var.z)) * scale);
    const v3 lo = m1 - dev, hi = m1 + dev;
    const v3 pmin = mk3(fminf(lo.x, center.x), fminf(lo.y, center.y), fminf(lo.z, center.z));
    const v3 pmax = mk3(fmaxf(hi.x, center.x), fmaxf(hi.y, center.y), fmaxf(hi.z, center.z));
    history       = clip_to_aabb(pmin, pmax, history);
    const v3 r    = invert_simple_tonemap(ycocg2rgb(lerp3(c, history, alpha)));
    out[o]        = make_float4(r.x, r.y, r.z, 1.0f);
}
}  // namespace

void launch_post_chain(hipStream_t stream, const PostChainArgs& a)
{
    const uint32_t W = a.width, H = a.height;
    const dim3     grid((W + 31) / 32, (H + 7) / 8), block(kBlock);
    const size_t   bytes = sizeof(float4) * (size_t)W * H;
    auto           img   = [&](const float4* p) { return Img{p, W, H}; };
    const uint32_t src = (a.frame_count + 1) % 2, dst = a.frame_count % 2;  // raytracing_system.cpp:1709-1710, 1754-1755
    uint32_t cg = (W * H + kBlock - 1) / kBlock;  // grid of the streaming (stencil-free) kernels
    if (cg > 4096) cg = 4096;
    if (cg == 0) cg = 1;
    auto mark = [&](int pass) {
        if (a.mark) a.mark(a.mark_user, pass);
    };
    mark(0);
    if (a.settings.gather || a.settings.denoise)
        hipLaunchKernelGGL(k_decode_normals, dim3(cg), block, 0, stream, a.normal_depth, a.normals, W * H);
    // SpatialGather (cpp:1541-1604); with lowres_indirect the input, the grid and indirect_temp are (W/2, H/2)
    const bool     up = a.settings.lowres_indirect != 0;
    const uint32_t IW = up ? W >> 1 : W, IH = up ? H >> 1 : H;
    const Img      indirect_in{a.indirect, IW, IH};
    const int      ox = (int)((a.frame_count % 4u) / 2u), oy = (int)((a.frame_count % 4u) % 2u);
    if (a.settings.gather && up)
        hipLaunchKernelGGL(k_gather<true>, dim3((IW + 31) / 32, (IH + 7) / 8), block, 0, stream, a.settings, indirect_in, img(a.normals),
                           a.indirect_temp, ox, oy);
    else if (a.settings.gather)
        hipLaunchKernelGGL(k_gather<false>, grid, block, 0, stream, a.settings, indirect_in, img(a.normals), a.indirect_temp, 0, 0);
    else
        (void)hipMemcpyAsync(a.indirect_temp, a.indirect, sizeof(float4) * (size_t)IW * IH, hipMemcpyDeviceToDevice, stream);
    // IntegrateTemporally (cpp:1283-1342)
    mark(1);
    hipLaunchKernelGGL(k_accumulate, grid, block, 0, stream, a.settings, a.frame_count, a.camera, a.prev_camera,
                       Img{a.indirect_temp, IW, IH}, img(a.normal_depth), img(a.indirect_history[src]), img(a.moments_history[src]), img(a.prev_normal_depth),
                       a.indirect_history[dst], a.moments_history[dst]);
    // Denoise (cpp:1437-1538)
    mark(2);
    if (a.settings.denoise)
    {
        hipLaunchKernelGGL(k_blur_disocclusion, grid, block, 0, stream, a.settings, img(a.indirect_history[dst]), img(a.normals),
                           img(a.moments_history[dst]), a.temp[0]);
        hipLaunchKernelGGL(k_blur, grid, block, 0, stream, a.settings, 1u, img(a.temp[0]), img(a.normals), a.temp[1]);
        hipLaunchKernelGGL(k_blur, grid, block, 0, stream, a.settings, 3u, img(a.temp[1]), img(a.normals), a.temp[0]);
        if (a.settings.eaw5)
        {
            hipLaunchKernelGGL(k_blur, grid, block, 0, stream, a.settings, 5u, img(a.temp[0]), img(a.normals), a.temp[1]);
            hipLaunchKernelGGL(k_blur, grid, block, 0, stream, a.settings, 7u, img(a.temp[1]), img(a.normals), a.temp[0]);
        }
    }
    else
        (void)hipMemcpyAsync(a.temp[0], a.indirect_history[dst], bytes, hipMemcpyDeviceToDevice, stream);
    // CombineIllumination (cpp:1400-1435)
    mark(3);
    hipLaunchKernelGGL(k_combine, dim3(cg), block, 0, stream, a.temp[0], a.albedo, a.direct, W * H);
    // ApplyTAA (cpp:1344-1398)
    mark(4);
    hipLaunchKernelGGL(k_taa, grid, block, 0, stream, a.settings, a.camera, a.prev_camera, img(a.temp[0]), img(a.normal_depth),
                       img(a.combined_history[src]), a.combined_history[dst]);
    // CopyGBuffer of the next frame (cpp:955-1009)
    (void)hipMemcpyAsync(a.prev_normal_depth, a.normal_depth, bytes, hipMemcpyDeviceToDevice, stream);
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
