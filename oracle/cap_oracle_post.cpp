/*
 * cap_oracle_post.cpp — CPU ORACLE (test infrastructure only, see cap_oracle.h) for the reference's reconstruction chain,
 * SURVEY.md 8f-1:  Gather -> Accumulate -> BlurDisocclusion -> Blur x2|x4 -> Combine -> TAA
 * (/root/reference/src/core/shaders/spatial_gather.hlsl, temporal_accumulation.hlsl, eaw_blur.hlsl,
 * combine_illumination.hlsl with eaw_edge_stopping.h, aabb.h, color_space.h, math_functions.h, utils.h, camera.h;
 * pass order and buffers: src/systems/raytracing_system.cpp:262-317, 1283-1604, 1700-1790).
 *
 * The reference's default configuration (raytracing_system.h:22-27: LOWRES_INDIRECT / UPSCALE2X off, CALCULATE_VARIANCE and
 * USE_VARIANCE on or off: OraclePostSettings::use_variance) and, with OraclePostSettings::lowres_indirect, the half-resolution one (UPSCALE2X Gather and Accumulate,
 * spatial_gather.hlsl:36-46, temporal_accumulation.hlsl:228-235, 307-313).  PARITY UNPINNED against the real renderer (no tests, no golden images; the
 * reference stores these buffers as RGBA16F, this build keeps fp32).  Stated choices where HLSL/D3D leave room:
 *   - uint(x) of a negative float saturates to 0; int(x) truncates; an out-of-bounds texture read returns 0;
 *   - rcp(x) = 1/x, lerp(a,b,t) = a + t*(b - a), exp/pow through the exp2/log2 polynomials of the arithmetic contract;
 *   - dot() with the contract's fma order; textures start zero-filled.
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>

#include "cap_oracle.h"

namespace
{
struct f2
{
    float x, y;
};
struct f3
{
    float x, y, z;
};
struct f4
{
    float x, y, z, w;
};
inline f3    mk(float x, float y, float z) { return f3{x, y, z}; }
inline f3    operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline f3    operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3    operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline f3    operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
inline f3    operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
inline float dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
inline f3    normalize(f3 v)
{
    float inv = 1.0f / sqrtf(dot(v, v));
    return v * inv;
}
inline float length(f3 v) { return sqrtf(dot(v, v)); }
inline f3    lerp3(f3 a, f3 b, float t) { return mk(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)); }
inline float lerp1(float a, float b, float t) { return a + t * (b - a); }
inline float frac(float x) { return x - floorf(x); }
inline uint32_t as_uint(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
inline float as_float(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline uint32_t sat_uint(float f) { return f > 0.0f ? (uint32_t)f : 0u; }  // uint(x), negative saturates to 0

constexpr float kEps = 1e-8f;  // math_functions.h:4

// exp2 / log2 of the arithmetic contract (same polynomials as cap_oracle.cpp)
float log2_contract(float x)
{
    uint32_t b = as_uint(x);
    int      e = (int)((b >> 23) & 0xffu) - 127;
    float    m = as_float((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f)
    {
        m *= 0.5f;
        e += 1;
    }
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float p = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 0.111111111111f, 0.142857142857f), 0.2f), 0.333333333333f), 1.0f);
    return fmaf((2.0f * s) * p, 1.44269504088896341f, (float)e);
}
float exp2_contract(float y)
{
    float n = floorf(y + 0.5f);
    float f = y - n;
    float p = 1.535336188319500e-4f;
    p       = fmaf(p, f, 1.339887440266574e-3f);
    p       = fmaf(p, f, 9.618437357674640e-3f);
    p       = fmaf(p, f, 5.550332471162809e-2f);
    p       = fmaf(p, f, 2.402264791363012e-1f);
    p       = fmaf(p, f, 6.931472028550421e-1f);
    p       = fmaf(p, f, 1.0f);
    return as_float(as_uint(p) + ((uint32_t)(int)n << 23));
}
// exp(x) for x <= 0 (edge-stopping weights); below 2^-125 the result is 0
float exp_neg(float x)
{
    float y = x * 1.44269504088896341f;
    if (!(y >= -125.0f)) return 0.0f;
    return exp2_contract(y);
}
// pow(x, s) for x in [0, 1], s > 0 (eaw_edge_stopping.h:4-7)
float pow01(float x, float s)
{
    if (!(x >= 1.17549435e-38f)) return 0.0f;
    float y = s * log2_contract(x);
    if (!(y >= -125.0f)) return 0.0f;
    return exp2_contract(y);
}

float luminance(f3 c) { return dot(c, mk(0.299f, 0.587f, 0.114f)); }  // math_functions.h:24-27, color_space.h:4-7

struct Image
{
    uint32_t        w = 0, h = 0;
    std::vector<f4> px;
    void            init(uint32_t ww, uint32_t hh) { w = ww, h = hh, px.assign((size_t)ww * hh, f4{0, 0, 0, 0}); }
    f4              load(uint32_t x, uint32_t y) const { return (x < w && y < h) ? px[(size_t)y * w + x] : f4{0, 0, 0, 0}; }
    f4              loadi(int x, int y) const { return (x >= 0 && y >= 0) ? load((uint32_t)x, (uint32_t)y) : f4{0, 0, 0, 0}; }
    void            store(uint32_t x, uint32_t y, f4 v) { if (x < w && y < h) px[(size_t)y * w + x] = v; }
};
inline f3 xyz(f4 v) { return mk(v.x, v.y, v.z); }

// utils.h:6-16
f2 uv_to_xy(f2 uv, uint32_t w, uint32_t h)
{
    float x = uv.x * (float)w, y = uv.y * (float)h;
    return f2{fminf(x, (float)(w - 1)), fminf(y, (float)(h - 1))};
}
f2 xy_to_uv(f2 xy, uint32_t w, uint32_t h)
{
    float u = xy.x / (float)w, v = xy.y / (float)h;
    return f2{fminf(fmaxf(u, 0.0f), 1.0f), fminf(fmaxf(v, 0.0f), 1.0f)};
}
// utils.h:20-35
f3 sample_bilinear(const Image& t, f2 uv)
{
    f2       xy = uv_to_xy(uv, t.w, t.h);
    float    fx = xy.x - 0.5f, fy = xy.y - 0.5f;
    uint32_t ux = sat_uint(floorf(fx)), uy = sat_uint(floorf(fy));
    float    wx = frac(fx), wy = frac(fy);
    f3 v00 = xyz(t.load(ux, uy)), v01 = xyz(t.load(ux, uy + 1)), v10 = xyz(t.load(ux + 1, uy)), v11 = xyz(t.load(ux + 1, uy + 1));
    return lerp3(lerp3(v00, v10, wx), lerp3(v01, v11, wx), wy);
}
// math_functions.h:60-77
float cubic(float x, float b, float c)
{
    float y = 0.0f, x2 = x * x, x3 = x * x * x;
    if (x < 1.0f)
        y = (12.0f - 9.0f * b - 6.0f * c) * x3 + (-18.0f + 12.0f * b + 6.0f * c) * x2 + (6.0f - 2.0f * b);
    else if (x <= 2.0f)
        y = (-b - 6.0f * c) * x3 + (6.0f * b + 30.0f * c) * x2 + (-12.0f * b - 48.0f * c) * x + (8.0f * b + 24.0f * c);
    return y / 6.0f;
}
// temporal_accumulation.hlsl:39-66
f3 resample_bicubic(const Image& t, f2 uv)
{
    f3    filtered = mk(0, 0, 0);
    f2    c        = uv_to_xy(uv, t.w, t.h);
    float tw       = 0.0f;
    for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j)
        {
            f2 cur = f2{c.x + (float)i, c.y + (float)j};
            bool offscreen = cur.x < 0.0f || cur.y < 0.0f || cur.x >= (float)t.w || cur.y >= (float)t.h;
            if (offscreen) continue;
            f3    value = sample_bilinear(t, xy_to_uv(cur, t.w, t.h));
            float dx = fabsf(cur.x - c.x), dy = fabsf(cur.y - c.y);
            float w = cubic(dx, 0.0f, 0.5f) * cubic(dy, 0.0f, 0.5f) * (1.0f / (1.0f + luminance(value)));
            filtered = filtered + value * w;
            tw += w;
        }
    return tw > 1e-5f ? filtered / tw : mk(0, 0, 0);
}

// math_functions.h:49-57
f3 oct_decode(float fx, float fy)
{
    fx = fx * 2.0f - 1.0f, fy = fy * 2.0f - 1.0f;
    f3    n = mk(fx, fy, 1.0f - fabsf(fx) - fabsf(fy));
    float t = fminf(fmaxf(-n.z, 0.0f), 1.0f);
    n.x += n.x >= 0.0f ? -t : t;
    n.y += n.y >= 0.0f ? -t : t;
    return normalize(n);
}
// eaw_edge_stopping.h
float normal_weight(f3 n0, f3 n1, float s) { return pow01(fmaxf(dot(n0, n1), 0.0f), s); }
float depth_weight(float dc, float dp, float s)
{
    float t = s == 0.0f ? 0.0f : (fabsf(dc - dp) / s);
    return exp_neg(-t);
}
float luma_weight(float lc, float lp, float s) { return exp_neg(-(fabsf(lc - lp) / s)); }

// color_space.h
f3 rgb2ycocg(f3 c) { return mk(c.x / 4.0f + c.y / 2.0f + c.z / 4.0f, c.x / 2.0f - c.z / 2.0f, -c.x / 4.0f + c.y / 2.0f - c.z / 4.0f); }
f3 ycocg2rgb(f3 c)
{
    auto cl = [](float v) { return fminf(fmaxf(v, 0.0f), 1.0f); };
    return mk(cl(c.x + c.y - c.z), cl(c.x + c.z), cl(c.x - c.y - c.z));
}
f3 simple_tonemap(f3 v) { return v / (1.0f + luminance(v)); }
f3 invert_simple_tonemap(f3 v) { return v / (1.0f - luminance(v)); }

// camera.h:8-37, 64-80
f2 image_plane_uv(const OracleCamera& cam, f3 position)
{
    f3    o = mk(cam.position[0], cam.position[1], cam.position[2]);
    f3    d = normalize(position - o);
    f3    n = normalize(mk(cam.forward[0], cam.forward[1], cam.forward[2]));
    f3    p = o + n * cam.focal_length;
    float t = dot(n, p - o) / dot(n, d);
    f3    ip = o + d * t;
    f3    ipd = ip - p;
    float u = dot(mk(cam.right[0], cam.right[1], cam.right[2]), ipd) / (0.5f * cam.sensor_size[0]);
    float v = dot(mk(cam.up[0], cam.up[1], cam.up[2]), ipd) / (0.5f * cam.sensor_size[1]);
    return f2{0.5f * u + 0.5f, 0.5f * v + 0.5f};
}
f3 reconstruct_world_position(const OracleCamera& cam, f2 uv, float depth)
{
    float cx = (uv.x - 0.5f) * cam.sensor_size[0], cy = (uv.y - 0.5f) * cam.sensor_size[1];
    f3    d  = normalize(mk(fmaf(cy, cam.up[0], fmaf(cx, cam.right[0], cam.focal_length * cam.forward[0])),
                            fmaf(cy, cam.up[1], fmaf(cx, cam.right[1], cam.focal_length * cam.forward[1])),
                            fmaf(cy, cam.up[2], fmaf(cx, cam.right[2], cam.focal_length * cam.forward[2]))));
    return mk(cam.position[0], cam.position[1], cam.position[2]) + d * depth;
}


// Rows of a pass on `g_post_threads` host threads (oracle_post_set_threads; default 1).  Every pass writes images it does not read and a
// pixel's value depends on no other output pixel, so the result does not depend on the thread count.
static int g_post_threads = 1;
template <class F>
static void post_rows(uint32_t H, F&& body)
{
    const int nt = g_post_threads < 1 ? 1 : g_post_threads;
    if (nt == 1 || H < 64u)
    {
        for (uint32_t y = 0; y < H; ++y) body(y);
        return;
    }
    std::vector<std::thread> pool;
    std::atomic<uint32_t>    next{0};
    for (int t = 0; t < nt; ++t)
        pool.emplace_back([&]() {
            for (;;)
            {
                const uint32_t y0 = next.fetch_add(8u);
                if (y0 >= H) return;
                for (uint32_t y = y0; y < y0 + 8u && y < H; ++y) body(y);
            }
        });
    for (auto& th : pool) th.join();
}

struct Chain
{
    uint32_t w, h;
    Image    indirect_history[2], moments_history[2], combined_history[2], prev_nd;
    Image    indirect_temp, temp[2];
};

// spatial_gather.hlsl:28-109
void gather(const OraclePostSettings& s, const Image& color, const Image& nd, Image& out)
{
    post_rows(out.h, [&](uint32_t y) {
        for (uint32_t x = 0; x < out.w; ++x)
        {
            f4    cg = nd.load(x, y);
            f3    cn = oct_decode(cg.x, cg.y);
            float cd = cg.w;
            f3    cc = xyz(color.load(x, y));
            if (cd < 1e-5f)
            {
                out.store(x, y, f4{cc.x, cc.y, cc.z, 0.0f});
                continue;
            }
            const float s_depth = cd * s.gather_depth_sigma, s_normal = s.gather_normal_sigma, s_luma = s.gather_luma_sigma;
            f3    filtered = mk(0, 0, 0);
            float total    = 0.0f;
            for (int dy = -3; dy <= 3; ++dy)
                for (int dx = -3; dx <= 3; ++dx)
                {
                    int sx = (int)x + dx, sy = (int)y + dy;
                    if (sx < 0 || sy < 0 || sx >= (int)out.w || sy >= (int)out.h) continue;
                    f3 c = xyz(color.loadi(sx, sy));
                    f4 g = nd.loadi(sx, sy);
                    if (g.w < 1e-5f) continue;
                    f3    n = oct_decode(g.x, g.y);
                    float len = sqrtf((float)(dx * dx + dy * dy));
                    float wgt = normal_weight(cn, n, s_normal) * depth_weight(cd, g.w, s_depth * len) * luma_weight(luminance(cc), luminance(c), s_luma);
                    filtered = filtered + c * wgt;
                    total += wgt;
                }
            f3 r = (total < kEps) ? cc : filtered / total;
            out.store(x, y, f4{r.x, r.y, r.z, 1.0f});
        }
    });
}

// spatial_gather.hlsl:28-109 with UPSCALE2X (:36-46, :83-87): the grid and `color` are half resolution, the G-buffer is read at
// (xy << 1) + sp_offset.  The host passes the FULL window size as g_constants.width/height in this mode too
// (raytracing_system.cpp:1562-1569), so the taps' bound test is against the full size; a tap beyond the half-resolution image
// reads colour 0 and a G-buffer texel outside the window, i.e. depth 0, and is skipped as background.
void gather_lowres(const OraclePostSettings& s, uint32_t frame_count, uint32_t full_w, uint32_t full_h, const Image& color, const Image& nd,
                   Image& out)
{
    const int ox = (int)((frame_count % 4) / 2), oy = (int)((frame_count % 4) % 2);
    post_rows(out.h, [&](uint32_t y) {
        for (uint32_t x = 0; x < out.w; ++x)
        {
            f4    cg = nd.loadi(((int)x << 1) + ox, ((int)y << 1) + oy);
            f3    cn = oct_decode(cg.x, cg.y);
            float cd = cg.w;
            f3    cc = xyz(color.load(x, y));
            if (cd < 1e-5f)
            {
                out.store(x, y, f4{cc.x, cc.y, cc.z, 0.0f});
                continue;
            }
            const float s_depth = cd * s.gather_depth_sigma, s_normal = s.gather_normal_sigma, s_luma = s.gather_luma_sigma;
            f3    filtered = mk(0, 0, 0);
            float total    = 0.0f;
            for (int dy = -3; dy <= 3; ++dy)
                for (int dx = -3; dx <= 3; ++dx)
                {
                    int sx = (int)x + dx, sy = (int)y + dy;
                    if (sx < 0 || sy < 0 || sx >= (int)full_w || sy >= (int)full_h) continue;
                    f3 c = xyz(color.loadi(sx, sy));
                    f4 g = nd.loadi((sx << 1) + ox, (sy << 1) + oy);
                    if (g.w < 1e-5f) continue;
                    f3    n = oct_decode(g.x, g.y);
                    float len = sqrtf((float)(dx * dx + dy * dy));
                    float wgt = normal_weight(cn, n, s_normal) * depth_weight(cd, g.w, s_depth * len) * luma_weight(luminance(cc), luminance(c), s_luma);
                    filtered = filtered + c * wgt;
                    total += wgt;
                }
            f3 r = (total < kEps) ? cc : filtered / total;
            out.store(x, y, f4{r.x, r.y, r.z, 1.0f});
        }
    });
}

float closest_depth(const Image& g, f2 xy)  // temporal_accumulation.hlsl:179-205
{
    float closest = g.loadi((int)xy.x, (int)xy.y).w;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
        {
            int tx = (int)xy.x + dx, ty = (int)xy.y + dy;
            if ((float)tx >= (float)g.w || (float)ty >= (float)g.h || tx < 0 || ty < 0) continue;
            f4 v = g.loadi(tx, ty);
            if (v.w != 0.0f && v.w < closest) closest = v.w;
        }
    return closest;
}

// temporal_accumulation.hlsl:213-325.  With UPSCALE2X (s.lowres_indirect) `color` is the half-resolution image: SampleColor uses its
// size (input_buffer_size, :228-235), and pixels that got no new sample this frame keep their history untouched (:307-313).
void accumulate(const OraclePostSettings& s, uint32_t frame_count, const OracleCamera& cam, const OracleCamera& prev_cam,
                const Image& color, const Image& nd, const Image& color_history, const Image& moments_history, const Image& prev_nd,
                Image& out_color, Image& out_moments)
{
    const uint32_t W = out_color.w, H = out_color.h;
    post_rows(H, [&](uint32_t y) {
        for (uint32_t x = 0; x < W; ++x)
        {
            f2 uv = f2{((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
            f4 g  = nd.load(x, y);
            auto reset = [&]() {
                f3    c = sample_bilinear(color, uv);
                float l = luminance(c);
                out_color.store(x, y, f4{c.x, c.y, c.z, 0.0f});
                out_moments.store(x, y, f4{l, l * l, 0.0f, 1.0f});
            };
            if (g.w < 1e-5f)
            {
                reset();
                continue;
            }
            f3 hit = reconstruct_world_position(cam, uv, g.w);
            f2 puv = image_plane_uv(prev_cam, hit);
            bool disocclusion = puv.x < 0.0f || puv.y < 0.0f || puv.x > 1.0f || puv.y > 1.0f || frame_count == 0;
            if (disocclusion)
            {
                reset();
                continue;
            }
            f2    pxy = uv_to_xy(puv, W, H);
            float cur_depth = length(hit - mk(prev_cam.position[0], prev_cam.position[1], prev_cam.position[2]));
            float prev_depth = closest_depth(prev_nd, pxy);
            if (fabsf(prev_depth - cur_depth) / cur_depth > 0.05f)
            {
                reset();
                continue;
            }
            float alpha   = s.temporal_upscale_feedback;
            f3    history = resample_bicubic(color_history, puv);
            f3    c       = sample_bilinear(color, uv);
            uint32_t hist_len = sat_uint(moments_history.load(sat_uint(floorf(pxy.x)), sat_uint(floorf(pxy.y))).w);
            if (hist_len < 256u)
            {
                float t = 1.0f / (float)(hist_len + 1);
                alpha   = fminf(alpha, 1.0f - t);
            }
            if (s.lowres_indirect && ((x % 2u) != (frame_count % 4) / 2 || (y % 2u) != (frame_count % 4) % 2))
            {
                alpha = 1.0f;
                hist_len -= 1;  // uint: a length of 0 wraps and the + 1 below brings it back to 0
            }
            f3    mh = resample_bicubic(moments_history, puv);
            float l  = luminance(c);
            float m0 = lerp1(l, mh.x, alpha), m1 = lerp1(l * l, mh.y, alpha);
            float variance = fabsf(m1 - m0 * m0);
            out_moments.store(x, y, f4{m0, m1, 0.0f, (float)(hist_len + 1)});
            f3 blended = lerp3(c, history, alpha);
            out_color.store(x, y, f4{blended.x, blended.y, blended.z, variance});
        }
    });
}

f3 remove_fireflies(f4 v) { return mk(fminf(v.x, 10.0f), fminf(v.y, 10.0f), fminf(v.z, 10.0f)); }  // eaw_blur.hlsl:30-33

// eaw_blur.hlsl:142-223
void blur_disocclusion(const OraclePostSettings& s, const Image& color, const Image& nd, const Image& moments, Image& out)
{
    post_rows(out.h, [&](uint32_t y) {
        for (uint32_t x = 0; x < out.w; ++x)
        {
            float hist = moments.load(x, y).w;
            f4    cg   = nd.load(x, y);
            f3    cn   = oct_decode(cg.x, cg.y);
            float cd   = cg.w;
            f4    cv   = color.load(x, y);
            f3    cc   = remove_fireflies(cv);
            float cvar = s.use_variance ? cv.w : 0.0f;  // eaw_blur.hlsl:160-165: center_variance stays 0 without USE_VARIANCE
            if (cd < 1e-5f || hist >= 8.0f)
            {
                out.store(x, y, f4{cc.x, cc.y, cc.z, cvar});
                continue;
            }
            const float s_depth = cd * s.eaw_depth_sigma, s_normal = s.eaw_normal_sigma, s_luma = s.eaw_luma_sigma;
            f3    filtered = mk(0, 0, 0);
            float fm0 = 0.0f, fm1 = 0.0f, total = 0.0f;
            for (int dy = -3; dy <= 3; ++dy)
                for (int dx = -3; dx <= 3; ++dx)
                {
                    int sx = (int)x + dx, sy = (int)y + dy;
                    if (sx < 0 || sy < 0 || sx >= (int)out.w || sy >= (int)out.h) continue;
                    f3 c = remove_fireflies(color.loadi(sx, sy));
                    f4 g = nd.loadi(sx, sy);
                    f4 m = moments.loadi(sx, sy);
                    if (g.w < 1e-5f) continue;
                    f3    n = oct_decode(g.x, g.y);
                    float len = sqrtf((float)(dx * dx + dy * dy));
                    float wgt = normal_weight(cn, n, s_normal) * depth_weight(cd, g.w, s_depth * len) * luma_weight(luminance(cc), luminance(c), s_luma);
                    filtered = filtered + c * wgt;
                    fm0 += wgt * m.x, fm1 += wgt * m.y;
                    total += wgt;
                }
            f3    r  = (total < kEps) ? cc : filtered / total;
            float m0 = (total < kEps) ? 0.0f : fm0 / total, m1 = (total < kEps) ? 0.0f : fm1 / total;
            float boost = 8.0f / hist;
            out.store(x, y, f4{r.x, r.y, r.z, boost * fabsf(m1 - m0 * m0)});
        }
    });
}

// eaw_blur.hlsl:48-137
void blur(const OraclePostSettings& s, uint32_t stride, const Image& color, const Image& nd, Image& out)
{
    const float kw[3] = {1.0f, 2.0f / 3.0f, 1.0f / 6.0f};
    post_rows(out.h, [&](uint32_t y) {
        for (uint32_t x = 0; x < out.w; ++x)
        {
            f4    cg   = nd.load(x, y);
            f3    cn   = oct_decode(cg.x, cg.y);
            float cd   = cg.w;
            f4    cv   = color.load(x, y);
            f3    cc   = remove_fireflies(cv);
            float cvar = s.use_variance ? cv.w : 0.0f;  // eaw_blur.hlsl:66-70
            if (cd < 1e-5f)
            {
                out.store(x, y, f4{cc.x, cc.y, cc.z, cvar});
                continue;
            }
            const float s_depth = cd * (float)stride * s.eaw_depth_sigma, s_normal = s.eaw_normal_sigma;
            const float s_luma = s.eaw_luma_sigma * sqrtf(fmaxf(0.0f, cvar + kEps));
            f3    filtered = mk(0, 0, 0);
            float fvar = 0.0f, total = 0.0f;
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx)
                {
                    int sx = (int)x + dx * (int)stride, sy = (int)y + dy * (int)stride;
                    if (sx < 0 || sy < 0 || sx >= (int)out.w || sy >= (int)out.h) continue;
                    f4 v = color.loadi(sx, sy);
                    f3 c = remove_fireflies(v);
                    f4 g = nd.loadi(sx, sy);
                    if (g.w < 1e-5f) continue;
                    f3    n  = oct_decode(g.x, g.y);
                    // eaw_blur.hlsl:110-118: without USE_VARIANCE both stay 1 and no variance is filtered
                    float lw = s.use_variance ? luma_weight(luminance(cc), luminance(c), s_luma) : 1.0f;
                    float hw = s.use_variance ? kw[dx < 0 ? -dx : dx] * kw[dy < 0 ? -dy : dy] : 1.0f;
                    float len = sqrtf((float)(dx * dx + dy * dy));
                    float wgt = normal_weight(cn, n, s_normal) * depth_weight(cd, g.w, s_depth * len);
                    float k   = wgt * hw * lw;
                    filtered = filtered + c * k;
                    total += k;
                    if (s.use_variance) fvar += hw * hw * wgt * wgt * lw * lw * v.w;
                }
            f3    r  = (total < kEps) ? cc : filtered / total;
            float rv = (total < kEps) ? cvar : fvar / (total * total);
            out.store(x, y, f4{r.x, r.y, r.z, rv});
        }
    });
}

// aabb.h:24-34
f3 clip_to_aabb(f3 pmin, f3 pmax, f3 p)
{
    f3 c = (pmin + pmax) * 0.5f, radius = (pmax - pmin) * 0.5f, dc = p - c;
    f3 clip = mk(dc.x / (radius.x + 1e-5f), dc.y / (radius.y + 1e-5f), dc.z / (radius.z + 1e-5f));
    float m = fmaxf(fmaxf(fabsf(clip.x), fabsf(clip.y)), fabsf(clip.z));
    return m > 1.0f ? c + dc / m : p;
}

// temporal_accumulation.hlsl:362-447
void taa(const OraclePostSettings& s, const OracleCamera& cam, const OracleCamera& prev_cam, const Image& color, const Image& nd,
         const Image& history_img, Image& out)
{
    const uint32_t W = out.w, H = out.h;
    post_rows(H, [&](uint32_t y) {
        for (uint32_t x = 0; x < W; ++x)
        {
            f2 uv = f2{((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
            f4 g  = nd.load(x, y);
            if (g.w < 1e-5f)
            {
                f3 c = sample_bilinear(color, uv);
                out.store(x, y, f4{c.x, c.y, c.z, 1.0f});
                continue;
            }
            f3 hit = reconstruct_world_position(cam, uv, g.w);
            f2 puv = image_plane_uv(prev_cam, hit);
            float vx = (puv.x - uv.x) * (float)W, vy = (puv.y - uv.y) * (float)H;
            float velocity = sqrtf(fmaf(vy, vy, vx * vx));
            bool  disocclusion = puv.x < 0.0f || puv.y < 0.0f || puv.x > 1.0f || puv.y > 1.0f;
            if (disocclusion)
            {
                f3 c = sample_bilinear(color, uv);
                out.store(x, y, f4{c.x, c.y, c.z, 1.0f});
                continue;
            }
            bool  is_static = velocity < 1e-3f;
            float alpha = is_static ? 0.98f : 0.6f, scale = is_static ? 5.0f : 0.75f;
            alpha = fminf(s.taa_feedback, alpha);
            f3 history = rgb2ycocg(simple_tonemap(resample_bicubic(history_img, puv)));
            f3 c       = rgb2ycocg(simple_tonemap(sample_bilinear(color, uv)));
            // CalculateNeighbourhoodColorAABB(gidx, dim, scale), :98-137
            f3 center = rgb2ycocg(simple_tonemap(sample_bilinear(color, xy_to_uv(f2{(float)x, (float)y}, W, H))));
            f3 m1 = mk(0, 0, 0), m2 = mk(0, 0, 0);
            for (int i = -2; i <= 2; ++i)
                for (int j = -2; j <= 2; ++j)
                {
                    int sx = (int)x + i, sy = (int)y + j;
                    sx = sx < 0 ? 0 : (sx > (int)W - 1 ? (int)W - 1 : sx);
                    sy = sy < 0 ? 0 : (sy > (int)H - 1 ? (int)H - 1 : sy);
                    f3 v = rgb2ycocg(simple_tonemap(sample_bilinear(color, xy_to_uv(f2{(float)sx, (float)sy}, W, H))));
                    m1 = m1 + v;
                    m2 = m2 + v * v;
                }
            const float inv_n = 1.0f / 25.0f;
            m1 = m1 * inv_n, m2 = m2 * inv_n;
            f3 var = m2 - m1 * m1;
            f3 dev = mk(sqrtf(fabsf(var.x)) * scale, sqrtf(fabsf(var.y)) * scale, sqrtf(fabsf(var.z)) * scale);
            f3 lo = m1 - dev, hi = m1 + dev;
            f3 pmin = mk(fminf(lo.x, center.x), fminf(lo.y, center.y), fminf(lo.z, center.z));
            f3 pmax = mk(fmaxf(hi.x, center.x), fmaxf(hi.y, center.y), fmaxf(hi.z, center.z));
            history = clip_to_aabb(pmin, pmax, history);
            f3 r = invert_simple_tonemap(ycocg2rgb(lerp3(c, history, alpha)));
            out.store(x, y, f4{r.x, r.y, r.z, 1.0f});
        }
    });
}

void from_floats(Image& img, const float* p)
{
    for (size_t i = 0; i < img.px.size(); ++i) img.px[i] = f4{p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]};
}
}  // namespace

extern "C" {

void* oracle_post_create(uint32_t w, uint32_t h)
{
    Chain* c = new Chain;
    c->w = w, c->h = h;
    for (int k = 0; k < 2; ++k) c->indirect_history[k].init(w, h), c->moments_history[k].init(w, h), c->combined_history[k].init(w, h), c->temp[k].init(w, h);
    c->prev_nd.init(w, h);
    c->indirect_temp.init(w, h);
    return c;
}

void oracle_post_destroy(void* h) { delete (Chain*)h; }

void oracle_post_set_threads(int n) { g_post_threads = n; }

int oracle_post_frame(void* handle, const OraclePostSettings* s, uint32_t frame_count, const OracleCamera* cam,
                      const OracleCamera* prev_cam, const float* indirect, const float* direct, const float* albedo,
                      const float* normal_depth, float* out)
{
    if (!handle || !s || !cam || !prev_cam || !indirect || !direct || !albedo || !normal_depth || !out) return 1;
    Chain&         c = *(Chain*)handle;
    const uint32_t W = c.w, H = c.h;
    const bool lowres = s->lowres_indirect != 0;
    if (lowres && ((W & 1u) || (H & 1u))) return 2;
    Image raw, dir, alb, nd;
    if (lowres)
        raw.init(W >> 1, H >> 1);  // output_indirect_ and indirect_temp_ are half resolution (raytracing_system.cpp:499-512)
    else
        raw.init(W, H);
    dir.init(W, H), alb.init(W, H), nd.init(W, H);
    from_floats(raw, indirect), from_floats(dir, direct), from_floats(alb, albedo), from_floats(nd, normal_depth);
    if (c.indirect_temp.w != raw.w || c.indirect_temp.h != raw.h) c.indirect_temp.init(raw.w, raw.h);
    const uint32_t src = (frame_count + 1) % 2, dst = frame_count % 2;  // raytracing_system.cpp:1709-1710, 1754-1755
    // SpatialGather (cpp:1541-1604)
    if (s->gather && lowres)
        gather_lowres(*s, frame_count, W, H, raw, nd, c.indirect_temp);
    else if (s->gather)
        gather(*s, raw, nd, c.indirect_temp);
    else
        c.indirect_temp = raw;
    // IntegrateTemporally (cpp:1283-1342)
    accumulate(*s, frame_count, *cam, *prev_cam, c.indirect_temp, nd, c.indirect_history[src], c.moments_history[src], c.prev_nd,
               c.indirect_history[dst], c.moments_history[dst]);
    // Denoise (cpp:1437-1538)
    if (s->denoise)
    {
        blur_disocclusion(*s, c.indirect_history[dst], nd, c.moments_history[dst], c.temp[0]);
        blur(*s, 1, c.temp[0], nd, c.temp[1]);
        blur(*s, 3, c.temp[1], nd, c.temp[0]);
        if (s->eaw5)
        {
            blur(*s, 5, c.temp[0], nd, c.temp[1]);
            blur(*s, 7, c.temp[1], nd, c.temp[0]);
        }
    }
    else
        c.temp[0] = c.indirect_history[dst];
    // CombineIllumination (combine_illumination.hlsl:16-40), in place; type = SettingsComponent::output (raytracing_system.cpp:1415)
    for (size_t i = 0; i < c.temp[0].px.size(); ++i)
    {
        f4 in = c.temp[0].px[i], a = alb.px[i], d = dir.px[i];
        switch (s->output)
        {
        case 0: c.temp[0].px[i] = f4{in.x * a.x + d.x, in.y * a.y + d.y, in.z * a.z + d.z, 1.0f * a.w + d.w}; break;  // :29 indirect = (xyz, 1)
        case 1: c.temp[0].px[i] = d; break;                                                                          // :32
        case 2: c.temp[0].px[i] = f4{in.x, in.y, in.z, 1.0f}; break;                                                 // :35
        case 3: c.temp[0].px[i] = f4{in.w, in.w, in.w, 1.0f}; break;                                                 // :38 .www
        default: return 3;
        }
    }
    // ApplyTAA (cpp:1344-1398)
    taa(*s, *cam, *prev_cam, c.temp[0], nd, c.combined_history[src], c.combined_history[dst]);
    memcpy(out, c.combined_history[dst].px.data(), sizeof(f4) * (size_t)W * H);
    // CopyGBuffer of the next frame (cpp:955-1009): prev_gbuffer_normal_depth <- gbuffer_normal_depth
    c.prev_nd = nd;
    return 0;
}

/* One pass of the chain on caller-supplied images (full resolution; w*h*4 floats each; unused inputs may be NULL): for known-answer
 * tests that pin a pass by itself against an independent evaluation of the shader text (tests/test_oracle_post_kat.py).
 * pass: 0 Gather(in0 = color, in1 = normal_depth) -> out0
 *       1 Accumulate(in0 = color, in1 = normal_depth, in2 = color history, in3 = moments history, in4 = previous normal_depth) ->
 *         out0 = color, out1 = moments          2 BlurDisocclusion(in0 = color, in1 = normal_depth, in2 = moments) -> out0
 *       3 Blur(stride = arg; in0 = color, in1 = normal_depth) -> out0      4 TAA(in0 = color, in1 = normal_depth, in2 = history) -> out0 */
int oracle_post_pass(int pass, const OraclePostSettings* s, uint32_t w, uint32_t h, uint32_t arg, const OracleCamera* cam,
                     const OracleCamera* prev_cam, const float* in0, const float* in1, const float* in2, const float* in3,
                     const float* in4, float* out0, float* out1)
{
    if (!s || !w || !h || !in0 || !in1 || !out0) return 1;
    Image a, b, c, d, e, o0, o1;
    for (Image* im : {&a, &b, &c, &d, &e, &o0, &o1}) im->init(w, h);
    from_floats(a, in0), from_floats(b, in1);
    if (in2) from_floats(c, in2);
    if (in3) from_floats(d, in3);
    if (in4) from_floats(e, in4);
    switch (pass)
    {
    case 0: gather(*s, a, b, o0); break;
    case 1:
        if (!cam || !prev_cam || !in2 || !in3 || !in4 || !out1) return 1;
        accumulate(*s, arg, *cam, *prev_cam, a, b, c, d, e, o0, o1);
        break;
    case 2:
        if (!in2) return 1;
        blur_disocclusion(*s, a, b, c, o0);
        break;
    case 3: blur(*s, arg, a, b, o0); break;
    case 4:
        if (!cam || !prev_cam || !in2) return 1;
        taa(*s, *cam, *prev_cam, a, b, c, o0);
        break;
    default: return 2;
    }
    memcpy(out0, o0.px.data(), sizeof(f4) * (size_t)w * h);
    if (out1) memcpy(out1, o1.px.data(), sizeof(f4) * (size_t)w * h);
    return 0;
}
}
