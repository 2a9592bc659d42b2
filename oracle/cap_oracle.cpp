/*
 * cap_oracle.cpp — CPU ORACLE: scalar fp32 restatement of the reference hot path.
 * TEST INFRASTRUCTURE ONLY (see cap_oracle.h).  PARITY UNPINNED at the TraceRay boundary.
 *
 * Every function cites the reference file:line (relative to /root/reference/src/core) it follows.
 *
 * Arithmetic contract (DESIGN.md "fp32 arithmetic contract"): IEEE-754 binary32, round to nearest,
 * no implicit contraction (built with -ffp-contract=off), fused multiply-add only where fmaf() is
 * written.  HLSL leaves the exact rounding of mad/dot/normalize/sin/cos/pow to the compiler and
 * driver, so the contract below is the build's own pinning of them; the HIP
 * path implements the same contract independently and must agree bit for bit.
 *
 *   dot(a,b)    = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   cross(a,b)  = ( fma(a.y,b.z, -(a.z*b.y)), fma(a.z,b.x, -(a.x*b.z)), fma(a.x,b.y, -(a.y*b.x)) )
 *   normalize(v)= v * (1 / sqrt(dot(v,v)))          length(v) = sqrt(dot(v,v))
 *   max/min     = IEEE maxNum/minNum (a NaN operand is dropped, as HLSL max/min do)
 *   sin/cos     = 3-term Cody-Waite reduction by pi/2 + Cephes single-precision minimax polynomials
 *   pow(x,2.2)  = exp2(2.2*log2(x)), log2 through the atanh series, exp2 through a degree-6 polynomial
 *   pow(x,0.5)  = sqrt(x)   (MapToHemisphere with e = 1; identical value for x >= 0 when correctly rounded)
 */
#include "cap_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace
{
constexpr uint32_t kInvalidId = ~0u;  // data_payload.h:5

struct f3
{
    float x, y, z;
};
inline f3    make3(float x, float y, float z) { return f3{x, y, z}; }
inline f3    operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline f3    operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3    operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline f3    operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
inline float dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
inline f3    cross(f3 a, f3 b)
{
    return f3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
inline float length(f3 v) { return sqrtf(dot(v, v)); }
inline f3    normalize(f3 v)
{
    float inv = 1.0f / sqrtf(dot(v, v));
    return v * inv;
}
// HLSL max/min drop a NaN operand; fmaxf/fminf have exactly that semantic.
inline float hmax(float a, float b) { return fmaxf(a, b); }
inline float hmin(float a, float b) { return fminf(a, b); }
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

constexpr float kPi    = 3.141592653589793238463f;  // sampling.h:4
constexpr float kInvPi = 1.0f / kPi;                // shading.h:16 (constant-folded in fp32)

// ---------------------------------------------------------------------------------------------
// Transcendentals of the arithmetic contract.
// ---------------------------------------------------------------------------------------------
void sincos_contract(float x, float* s, float* c)
{
    // valid for |x| < ~100 (reference arguments are in [0, 2*pi]: sampling.h:125-126, lighting.h:22-25)
    const float kTwoOverPi = 0.636619772367581343f;
    const float DP1 = 1.5703125f, DP2 = 4.837512969970703125e-4f, DP3 = 7.54978995489188216e-8f;
    float       kf = floorf(x * kTwoOverPi + 0.5f);
    int         k  = (int)kf;
    float       a  = fmaf(-kf, DP1, x);
    a              = fmaf(-kf, DP2, a);
    a              = fmaf(-kf, DP3, a);
    float z        = a * a;
    // sin(a) = a + a*z*(S1 + z*(S2 + z*S3));  cos(a) = 1 - z/2 + z*z*(C1 + z*(C2 + z*C3))
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sp = fmaf(ps * z, a, a);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cp = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    switch (k & 3)
    {
    case 0: *s = sp; *c = cp; break;
    case 1: *s = cp; *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
    }
}

float log2_contract(float x)  // x normal, > 0
{
    uint32_t b = as_uint(x);
    int      e = (int)((b >> 23) & 0xffu) - 127;
    float    m = as_float((b & 0x007fffffu) | 0x3f800000u);  // [1,2)
    if (m > 1.41421356237f)
    {
        m *= 0.5f;
        e += 1;
    }
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float p = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 0.111111111111f, 0.142857142857f), 0.2f), 0.333333333333f), 1.0f);
    float ln_m = (2.0f * s) * p;
    return fmaf(ln_m, 1.44269504088896341f, (float)e);
}

float exp2_contract(float y)  // y in [-126, 126]
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

// scene.h:58  kd = pow(kd, 2.2f)  (base in [0,1])
float pow22_contract(float x)
{
    if (!(x >= 1.17549435e-38f)) return 0.0f;  // 0, denormals, negatives, NaN
    float y = 2.2f * log2_contract(x);
    if (y < -125.0f) return 0.0f;
    return exp2_contract(y);
}

// ---------------------------------------------------------------------------------------------
// sampling.h
// ---------------------------------------------------------------------------------------------
// sampling.h:143-155
void halton23(uint32_t frame_count, float out[2])
{
    static const double pts[8][2] = {{0.5, 0.3333333333333333},   {0.25, 0.6666666666666666},
                                     {0.75, 0.1111111111111111},  {0.125, 0.4444444444444444},
                                     {0.625, 0.7777777777777777}, {0.375, 0.2222222222222222},
                                     {0.875, 0.5555555555555556}, {0.0625, 0.8888888888888888}};
    out[0] = (float)pts[frame_count % 8][0];
    out[1] = (float)pts[frame_count % 8][1];
}

// sampling.h:37-46
uint32_t wang_hash(uint32_t x, uint32_t y)
{
    const uint32_t M = 1664525u, C = 1013904223u;
    uint32_t       seed = (x * M + y + C) * M;
    seed ^= (seed >> 11u);
    seed ^= (seed << 7u) & 0x9d2c5680u;
    seed ^= (seed << 15u) & 0xefc60000u;
    seed ^= (seed >> 18u);
    return seed;
}

// sampling.h:13-23.  The texture is RGBA8 UNORM read with Load(): channel value = byte / 255.
void bluenoise4x4(const uint8_t* tex, uint32_t x, uint32_t y, uint32_t count, float out[2])
{
    uint32_t px = (count % 16) % 4;
    uint32_t py = (count % 16) / 4;
    uint32_t sx = (x * 4 + px) % 256;
    uint32_t sy = (y * 4 + py) % 256;
    const uint8_t* t = tex + 4 * (sy * 256 + sx);
    float vx = (float)t[0] / 255.0f;
    float vy = (float)t[1] / 255.0f;
    float k  = 0.61803398875f * (float)(count / 16);
    out[0]   = frac(vx + k);
    out[1]   = frac(vy + k);
}

// sampling.h:91-111
f3 ortho_vector(f3 n)
{
    f3 p;
    if (fabsf(n.z) > 0.0f)
    {
        float k = sqrtf(fmaf(n.z, n.z, n.y * n.y));  // length(n.yz)
        p.x     = 0.0f;
        p.y     = -n.z / k;
        p.z     = n.y / k;
    }
    else
    {
        float k = sqrtf(fmaf(n.y, n.y, n.x * n.x));  // length(n.xy)
        p.x     = n.y / k;
        p.y     = -n.x / k;
        p.z     = 0.0f;
    }
    return p;
}

// sampling.h:113-132 with e = 1 (shading.h:26): pow(1 - r2, 1/(e+1)) == sqrt(1 - r2).
f3 map_to_hemisphere(const float s[2], f3 n)
{
    f3 u = ortho_vector(n);
    f3 v = cross(u, n);
    u    = cross(n, v);
    float r1 = s[0], r2 = s[1];
    float sin_psi, cos_psi;
    sincos_contract((2.0f * kPi) * r1, &sin_psi, &cos_psi);
    float cos_theta = sqrtf(1.0f - r2);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float a = sin_theta * cos_psi;
    float b = sin_theta * sin_psi;
    f3    d = make3(fmaf(n.x, cos_theta, fmaf(v.x, b, u.x * a)), fmaf(n.y, cos_theta, fmaf(v.y, b, u.y * a)),
                    fmaf(n.z, cos_theta, fmaf(v.z, b, u.z * a)));
    return normalize(d);
}

// ---------------------------------------------------------------------------------------------
// lighting.h / shading.h
// ---------------------------------------------------------------------------------------------
struct LightSample
{
    f3 direction, intensity;
};
// lighting.h:20-33 (note the literal 3.14, evaluated in fp32 as the HLSL does)
LightSample directional_light(uint32_t count)
{
    float t = 2.0f * 3.14f * (float)(count % 4096) / 4096.0f;
    float st, ct;
    sincos_contract(t, &st, &ct);
    float       ly = 100.0f, lx = 40.0f * st, lz = 40.0f * ct;
    LightSample ls;
    ls.direction = normalize(make3(lx, ly, lz));
    ls.intensity = make3(1.0f * (2.0f * 14.0f + 0.0f), 1.0f * (2.0f * 12.0f + 0.0f), 1.0f * (2.0f * 10.0f + (2.0f + 2.0f * ct)));
    return ls;
}

// math_functions.h:36-47
void oct_encode(f3 n, float out[2])
{
    float s = fabsf(n.x) + fabsf(n.y) + fabsf(n.z);
    n       = make3(n.x / s, n.y / s, n.z / s);
    float ox = n.x, oy = n.y;
    if (!(n.z >= 0.0f))
    {
        ox = (1.0f - fabsf(n.y)) * (n.x >= 0.0f ? 1.0f : -1.0f);
        oy = (1.0f - fabsf(n.x)) * (n.y >= 0.0f ? 1.0f : -1.0f);
    }
    out[0] = ox * 0.5f + 0.5f;
    out[1] = oy * 0.5f + 0.5f;
}

// ---------------------------------------------------------------------------------------------
// Ray / triangle intersection (replaces the DXR fixed-function unit; contract in DESIGN.md):
// two-sided Moller-Trumbore evaluated in the determinant-scaled domain around a per-triangle plane
// normal, one division per accepted candidate; a hit needs tmin < t < tmax (DXR triangle rule);
// barycentrics (u,v) weight v1, v2.
// ---------------------------------------------------------------------------------------------
struct Tri
{
    f3       v0, e1, e2, n;  // n = cross(e1, e2), evaluated once per triangle
    uint32_t inst, prim;
};
struct Hit
{
    float    t, u, v;
    uint32_t gid;  // global triangle index; ~0u = miss
};

// Reciprocal of the intersection contract: bit-trick seed + three Newton-Raphson steps, each x <- x * fma(-a, x, 2).
// Relative error <= 1e-7 for normal a > 0; fully specified by IEEE operations, so host and device agree bit for bit
// (a hardware reciprocal estimate would not be reproducible on the CPU).
inline float rcp_contract(float a)
{
    float x = as_float(0x7EF311C7u - as_uint(a));
    x       = x * fmaf(-a, x, 2.0f);
    x       = x * fmaf(-a, x, 2.0f);
    x       = x * fmaf(-a, x, 2.0f);
    return x;
}

inline bool intersect_tri(f3 o, f3 d, const Tri& tr, float tmin, float tmax, float* t, float* u, float* v)
{
    // scalar triple products of Moller-Trumbore regrouped around the precomputed plane normal n = e1 x e2 and
    // q = tvec x d:  det = e1.(d x e2) = -d.n,  U = tvec.(d x e2) = e2.q,  V = d.(tvec x e1) = -e1.q,  T = e2.(tvec x e1) = tvec.n
    f3    tvec = o - tr.v0;
    f3    q    = cross(tvec, d);
    float det  = -dot(d, tr.n);
    float U = dot(tr.e2, q), V = -dot(tr.e1, q), T = dot(tvec, tr.n);
    if (det < 0.0f)
    {
        U = -U, V = -V, T = -T, det = -det;
    }
    if (!(det > 0.0f)) return false;
    if (!(U >= 0.0f && V >= 0.0f && U + V <= det)) return false;
    float inv = rcp_contract(det);
    float tt  = T * inv;
    if (!(tt > tmin && tt < tmax)) return false;
    *t = tt, *u = U * inv, *v = V * inv;
    return true;
}

// Occlusion query (lighting.h:48-55, ACCEPT_FIRST_HIT_AND_END_SEARCH: only "is there a hit" is observable).  Same
// determinant-scaled quantities; the open interval test is made in the scaled domain, tmin*det < T < tmax*det, which
// needs no division (the two forms differ only when T/det rounds across an interval end).
inline bool occludes_tri(f3 o, f3 d, const Tri& tr, float tmin, float tmax)
{
    f3    tvec = o - tr.v0;
    f3    q    = cross(tvec, d);
    float det  = -dot(d, tr.n);
    float U = dot(tr.e2, q), V = -dot(tr.e1, q), T = dot(tvec, tr.n);
    if (det < 0.0f)
    {
        U = -U, V = -V, T = -T, det = -det;
    }
    if (!(det > 0.0f)) return false;
    if (!(U >= 0.0f && V >= 0.0f && U + V <= det)) return false;
    return T > tmin * det && T < tmax * det;
}

// ---------------------------------------------------------------------------------------------
// Scene: pooled geometry (asset_load_system.cpp:162-255) flattened to one triangle list.  The
// reference's TLAS has one identity-transform instance per mesh with InstanceID = mesh.index
// (tlas_system.cpp:40-58), so (instance, primitive) == (mesh, triangle-in-mesh).
// ---------------------------------------------------------------------------------------------
struct BvhNode
{
    float    lo[3], hi[3];
    uint32_t left, right;   // children, or for a leaf: first, count | 0x80000000
};

struct Scene
{
    std::vector<float>          positions, normals, texcoords;
    std::vector<uint32_t>       indices;
    std::vector<OracleMesh>     meshes;
    std::vector<OracleTexture>  textures;
    std::vector<std::vector<uint8_t>> texture_data;
    std::vector<OracleMaterial> materials;
    std::vector<Tri>            tris;       // global triangle order = mesh order, then primitive order
    std::vector<uint32_t>       bvh_order;  // triangle ids in leaf order
    std::vector<BvhNode>        nodes;
    // EXT: emissive triangles in global triangle order, float prefix sums of their areas, total emissive area
    std::vector<uint32_t>       light_tris;
    std::vector<float>          light_cdf;
    float                       light_area = 0.0f;
};

void build_bvh(Scene& sc)
{
    size_t n = sc.tris.size();
    sc.bvh_order.resize(n);
    for (size_t i = 0; i < n; ++i) sc.bvh_order[i] = (uint32_t)i;
    sc.nodes.clear();
    if (n == 0) return;
    std::vector<f3> lo(n), hi(n), cen(n);
    for (size_t i = 0; i < n; ++i)
    {
        const Tri& t = sc.tris[i];
        f3 a = t.v0, b = t.v0 + t.e1, c = t.v0 + t.e2;
        // pad so the box certainly contains every point the fp32 intersection can report
        lo[i]  = make3(hmin(a.x, hmin(b.x, c.x)), hmin(a.y, hmin(b.y, c.y)), hmin(a.z, hmin(b.z, c.z)));
        hi[i]  = make3(hmax(a.x, hmax(b.x, c.x)), hmax(a.y, hmax(b.y, c.y)), hmax(a.z, hmax(b.z, c.z)));
        cen[i] = (lo[i] + hi[i]) * 0.5f;
    }
    struct Work
    {
        uint32_t node, first, count;
    };
    sc.nodes.push_back(BvhNode{});
    std::vector<Work> stack{{0, 0, (uint32_t)n}};
    while (!stack.empty())
    {
        Work w = stack.back();
        stack.pop_back();
        float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = w.first; i < w.first + w.count; ++i)
        {
            uint32_t id = sc.bvh_order[i];
            const float l[3] = {lo[id].x, lo[id].y, lo[id].z}, h[3] = {hi[id].x, hi[id].y, hi[id].z};
            const float c[3] = {cen[id].x, cen[id].y, cen[id].z};
            for (int k = 0; k < 3; ++k)
            {
                blo[k] = std::min(blo[k], l[k]), bhi[k] = std::max(bhi[k], h[k]);
                clo[k] = std::min(clo[k], c[k]), chi[k] = std::max(chi[k], c[k]);
            }
        }
        BvhNode& nd = sc.nodes[w.node];
        for (int k = 0; k < 3; ++k)
        {
            float pad = 1e-5f * std::max(1.0f, std::max(fabsf(blo[k]), fabsf(bhi[k])));
            nd.lo[k] = blo[k] - pad, nd.hi[k] = bhi[k] + pad;
        }
        int   axis = 0;
        float ext  = chi[0] - clo[0];
        for (int k = 1; k < 3; ++k)
            if (chi[k] - clo[k] > ext) ext = chi[k] - clo[k], axis = k;
        if (w.count <= 4 || !(ext > 0.0f))
        {
            nd.left = w.first, nd.right = w.count | 0x80000000u;
            continue;
        }
        uint32_t mid = w.first + w.count / 2;
        auto     key = [&](uint32_t id) { return axis == 0 ? cen[id].x : axis == 1 ? cen[id].y : cen[id].z; };
        std::nth_element(sc.bvh_order.begin() + w.first, sc.bvh_order.begin() + mid,
                         sc.bvh_order.begin() + w.first + w.count,
                         [&](uint32_t a, uint32_t b) { return key(a) < key(b) || (key(a) == key(b) && a < b); });
        uint32_t l = (uint32_t)sc.nodes.size();
        sc.nodes.push_back(BvhNode{});
        sc.nodes.push_back(BvhNode{});
        sc.nodes[w.node].left = l, sc.nodes[w.node].right = l + 1;
        stack.push_back({l, w.first, mid - w.first});
        stack.push_back({l + 1, mid, w.first + w.count - mid});
    }
}

// Conservative slab test: may accept a box the ray misses, never rejects one it hits within [tmin, tmax].
inline bool hit_box(const BvhNode& nd, f3 o, f3 inv, float tmin, float tmax)
{
    float t0x = (nd.lo[0] - o.x) * inv.x, t1x = (nd.hi[0] - o.x) * inv.x;
    float t0y = (nd.lo[1] - o.y) * inv.y, t1y = (nd.hi[1] - o.y) * inv.y;
    float t0z = (nd.lo[2] - o.z) * inv.z, t1z = (nd.hi[2] - o.z) * inv.z;
    float tn  = hmax(hmax(hmin(t0x, t1x), hmin(t0y, t1y)), hmax(hmin(t0z, t1z), tmin));
    float tf  = hmin(hmin(hmax(t0x, t1x), hmax(t0y, t1y)), hmin(hmax(t0z, t1z), tmax));
    return tn <= tf * 1.0000004f;
}

// Closest hit: minimum t over all triangles; equal t resolved towards the lower global triangle id, so the
// answer does not depend on the order triangles are visited (brute force == any BVH).
Hit trace_closest(const Scene& sc, f3 o, f3 d, float tmin, float tmax, bool use_bvh)
{
    Hit best{tmax, 0.0f, 0.0f, kInvalidId};
    auto test = [&](uint32_t id) {
        float t, u, v;
        if (intersect_tri(o, d, sc.tris[id], tmin, tmax, &t, &u, &v))
            if (t < best.t || (t == best.t && id < best.gid)) best = Hit{t, u, v, id};
    };
    if (!use_bvh || sc.nodes.empty())
    {
        for (uint32_t i = 0; i < sc.tris.size(); ++i) test(i);
        return best;
    }
    f3       inv = make3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t stack[128];
    int      sp = 0;
    stack[sp++] = 0;
    while (sp)
    {
        const BvhNode& nd = sc.nodes[stack[--sp]];
        if (!hit_box(nd, o, inv, tmin, best.t)) continue;
        if (nd.right & 0x80000000u)
        {
            uint32_t cnt = nd.right & 0x7fffffffu;
            for (uint32_t i = 0; i < cnt; ++i) test(sc.bvh_order[nd.left + i]);
        }
        else
        {
            stack[sp++] = nd.left;
            stack[sp++] = nd.right;
        }
    }
    return best;
}

// Any hit (lighting.h:48-55: FORCE_OPAQUE | ACCEPT_FIRST_HIT_AND_END_SEARCH): does any triangle satisfy tmin < t < tmax.
bool trace_any(const Scene& sc, f3 o, f3 d, float tmin, float tmax, bool use_bvh)
{
    float t, u, v;
    if (!use_bvh || sc.nodes.empty())
    {
        for (uint32_t i = 0; i < sc.tris.size(); ++i)
            if (occludes_tri(o, d, sc.tris[i], tmin, tmax)) return true;
        return false;
    }
    f3       inv = make3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t stack[128];
    int      sp = 0;
    stack[sp++] = 0;
    while (sp)
    {
        const BvhNode& nd = sc.nodes[stack[--sp]];
        if (!hit_box(nd, o, inv, tmin, tmax)) continue;
        if (nd.right & 0x80000000u)
        {
            uint32_t cnt = nd.right & 0x7fffffffu;
            for (uint32_t i = 0; i < cnt; ++i)
                if (occludes_tri(o, d, sc.tris[sc.bvh_order[nd.left + i]], tmin, tmax)) return true;
        }
        else
        {
            stack[sp++] = nd.left;
            stack[sp++] = nd.right;
        }
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// scene.h
// ---------------------------------------------------------------------------------------------
// scene.h:5-50
void interpolate_attributes(const Scene& sc, uint32_t instance, uint32_t prim, float u, float v, f3* p, f3* n,
                            float tx[2])
{
    const OracleMesh& mesh = sc.meshes[instance];
    uint32_t io = mesh.first_index_offset;
    uint32_t i0 = sc.indices[io + 3 * prim + 0], i1 = sc.indices[io + 3 * prim + 1], i2 = sc.indices[io + 3 * prim + 2];
    uint32_t vo = mesh.first_vertex_offset * 3;
    const float* P = sc.positions.data();
    const float* N = sc.normals.data();
    f3 v0 = make3(P[vo + 3 * i0], P[vo + 3 * i0 + 1], P[vo + 3 * i0 + 2]);
    f3 v1 = make3(P[vo + 3 * i1], P[vo + 3 * i1 + 1], P[vo + 3 * i1 + 2]);
    f3 v2 = make3(P[vo + 3 * i2], P[vo + 3 * i2 + 1], P[vo + 3 * i2 + 2]);
    f3 n0 = make3(N[vo + 3 * i0], N[vo + 3 * i0 + 1], N[vo + 3 * i0 + 2]);
    f3 n1 = make3(N[vo + 3 * i1], N[vo + 3 * i1 + 1], N[vo + 3 * i1 + 2]);
    f3 n2 = make3(N[vo + 3 * i2], N[vo + 3 * i2 + 1], N[vo + 3 * i2 + 2]);
    vo    = mesh.first_vertex_offset;
    const float* T = sc.texcoords.data();
    float w = (1.0f - u) - v;
    // a*w + b*u + c*v evaluated as fma(c, v, fma(b, u, a*w))
    auto mix = [&](float a, float b, float c) { return fmaf(c, v, fmaf(b, u, a * w)); };
    *n    = normalize(make3(mix(n0.x, n1.x, n2.x), mix(n0.y, n1.y, n2.y), mix(n0.z, n1.z, n2.z)));
    *p    = make3(mix(v0.x, v1.x, v2.x), mix(v0.y, v1.y, v2.y), mix(v0.z, v1.z, v2.z));
    tx[0] = mix(T[2 * (vo + i0)], T[2 * (vo + i1)], T[2 * (vo + i2)]);
    tx[1] = mix(T[2 * (vo + i0) + 1], T[2 * (vo + i1) + 1], T[2 * (vo + i2) + 1]);
}

// SampleLevel(g_sampler, tx, 0) with D3D12_FILTER_MIN_MAG_MIP_LINEAR and the static sampler's default WRAP
// addressing (raytracing_system.cpp:377, d3dx12.h:943-944) on an RGBA8 texture: texel centres at (i+0.5)/W.
void sample_texture(const OracleTexture& tex, float u, float v, float out[3])
{
    float fx = fmaf(u, (float)tex.width, -0.5f), fy = fmaf(v, (float)tex.height, -0.5f);
    float x0f = floorf(fx), y0f = floorf(fy);
    float wx = fx - x0f, wy = fy - y0f;
    auto  wrap = [](float f, uint32_t n) {
        float m = f - floorf(f / (float)n) * (float)n;  // [0, n)
        int   i = (int)m;
        if (i < 0) i = 0;
        if ((uint32_t)i >= n) i = 0;
        return (uint32_t)i;
    };
    uint32_t x0 = wrap(x0f, tex.width), y0 = wrap(y0f, tex.height);
    uint32_t x1 = (x0 + 1 == tex.width) ? 0 : x0 + 1, y1 = (y0 + 1 == tex.height) ? 0 : y0 + 1;
    for (int c = 0; c < 3; ++c)
    {
        auto  tx  = [&](uint32_t x, uint32_t y) { return (float)tex.rgba8[4 * (y * tex.width + x) + c] / 255.0f; };
        float top = fmaf(tx(x1, y0) - tx(x0, y0), wx, tx(x0, y0));
        float bot = fmaf(tx(x1, y1) - tx(x0, y1), wx, tx(x0, y1));
        out[c]    = fmaf(bot - top, wy, top);
    }
}

// scene.h:52-61
f3 get_material(const Scene& sc, uint32_t instance, const float tx_in[2])
{
    const OracleMesh& mesh = sc.meshes[instance];
    float kd[3];
    if (mesh.texture_index == kInvalidId || mesh.texture_index >= sc.textures.size())
        kd[0] = kd[1] = kd[2] = 0.75f;
    else
        sample_texture(sc.textures[mesh.texture_index], tx_in[0], 1.0f - tx_in[1], kd);
    return make3(pow22_contract(kd[0]), pow22_contract(kd[1]), pow22_contract(kd[2]));
}

// ---------------------------------------------------------------------------------------------
// camera.h:39-63
// ---------------------------------------------------------------------------------------------
void create_primary_ray(const OracleCamera& cam, uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t frame_count,
                        f3* origin, f3* dir)
{
    float s[2];
    halton23(frame_count, s);
    float ix = ((float)x + s[0]) / (float)w, iy = ((float)y + s[1]) / (float)h;
    float cx = (ix - 0.5f) * cam.sensor_size[0], cy = (iy - 0.5f) * cam.sensor_size[1];
    f3    d  = make3(fmaf(cy, cam.up[0], fmaf(cx, cam.right[0], cam.focal_length * cam.forward[0])),
                     fmaf(cy, cam.up[1], fmaf(cx, cam.right[1], cam.focal_length * cam.forward[1])),
                     fmaf(cy, cam.up[2], fmaf(cx, cam.right[2], cam.focal_length * cam.forward[2])));
    *dir     = normalize(d);
    *origin  = make3(cam.position[0], cam.position[1], cam.position[2]);
}

// ---------------------------------------------------------------------------------------------
// One pixel, one frame: rt_primary_visibility.hlsl:35-49, rt_direct_lighting.hlsl:38-83,
// rt_indirect.hlsl:46-177 (GBUFFER_FEEDBACK and LOWRES_INDIRECT off).
// ---------------------------------------------------------------------------------------------
struct PixelOut
{
    float geo[4], direct[4], albedo[4], nd[4], indirect[4];
};

const f3 kSky = {0.7f, 0.7f, 0.85f};  // rt_direct_lighting.hlsl:55, rt_indirect.hlsl:97

// lighting.h:35-61.  Unshadowed value first; the shadow ray is only traced when that value is non-zero
// (a zero contribution is the same image either way; rays are counted as traced).
f3 direct_illumination(const Scene& sc, const LightSample& ls, f3 p, f3 n, f3 kd, bool use_bvh, uint64_t* shadow_rays)
{
    float ndl = hmax(0.0f, dot(n, ls.direction));
    f3    c   = ((ls.intensity * kd) * kInvPi) * ndl;
    if (!(c.x != 0.0f || c.y != 0.0f || c.z != 0.0f)) return make3(0, 0, 0);
    ++*shadow_rays;
    if (trace_any(sc, p, ls.direction, 0.0001f, 100000.0f, use_bvh)) return make3(0, 0, 0);
    return c;
}

// ---- G-buffer feedback inputs (rt_indirect.hlsl:23, 34-35, 116-145) and the helpers the branch calls ----
struct Feedback
{
    OracleCamera prev_cam;
    const float* prev_normal_depth;  // W*H*4, the previous frame's gbuffer_normal_depth (CopyGBuffer, raytracing_system.cpp:955-1009)
    const float* color_history;      // W*H*4, combined_history[(frame_count + 1) % 2] = the previous frame's TAA output
};
struct f2
{
    float x, y;
};
// camera.h:8-37 CalculateImagePlaneUV
f2 image_plane_uv(const OracleCamera& cam, f3 position)
{
    f3    o = make3(cam.position[0], cam.position[1], cam.position[2]);
    f3    d = normalize(position - o);
    f3    n = normalize(make3(cam.forward[0], cam.forward[1], cam.forward[2]));
    f3    p = o + n * cam.focal_length;
    float t = dot(n, p - o) / dot(n, d);
    f3    ip = o + d * t;
    f3    ipd = ip - p;
    float u = dot(make3(cam.right[0], cam.right[1], cam.right[2]), ipd) / (0.5f * cam.sensor_size[0]);
    float v = dot(make3(cam.up[0], cam.up[1], cam.up[2]), ipd) / (0.5f * cam.sensor_size[1]);
    return f2{0.5f * u + 0.5f, 0.5f * v + 0.5f};
}
// utils.h:6-10
f2 uv_to_xy(f2 uv, uint32_t w, uint32_t h) { return f2{hmin(uv.x * (float)w, (float)(w - 1)), hmin(uv.y * (float)h, (float)(h - 1))}; }
f3 image_load3(const float* img, uint32_t w, uint32_t h, uint32_t x, uint32_t y)  // out-of-bounds reads return 0
{
    if (x >= w || y >= h) return make3(0, 0, 0);
    const float* p = img + 4 * ((size_t)y * w + x);
    return make3(p[0], p[1], p[2]);
}
// utils.h:20-35 SampleBilinear; uint(x) of a negative float saturates to 0, lerp(a,b,t) = a + t*(b - a)
f3 sample_bilinear(const float* img, uint32_t w, uint32_t h, f2 uv)
{
    f2       xy = uv_to_xy(uv, w, h);
    float    fx = xy.x - 0.5f, fy = xy.y - 0.5f;
    float    flx = floorf(fx), fly = floorf(fy);
    uint32_t ux = flx > 0.0f ? (uint32_t)flx : 0u, uy = fly > 0.0f ? (uint32_t)fly : 0u;
    float    wx = frac(fx), wy = frac(fy);
    f3 v00 = image_load3(img, w, h, ux, uy), v01 = image_load3(img, w, h, ux, uy + 1), v10 = image_load3(img, w, h, ux + 1, uy),
       v11 = image_load3(img, w, h, ux + 1, uy + 1);
    auto lerp3 = [](f3 a, f3 b, float t) { return make3(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)); };
    return lerp3(lerp3(v00, v10, wx), lerp3(v01, v11, wx), wy);
}

void shade_pixel(const Scene& sc, const OracleCamera& cam, const uint8_t* bn, uint32_t x, uint32_t y, uint32_t w,
                 uint32_t h, uint32_t frame_count, uint32_t num_bounces, bool use_bvh, PixelOut* o, uint64_t rays[3],
                 const Feedback* fb = nullptr, bool lowres = false)
{
    f3 org, dir;
    create_primary_ray(cam, x, y, w, h, frame_count, &org, &dir);
    Hit hit = trace_closest(sc, org, dir, 0.0f, 1e6f, use_bvh);  // camera.h:59-60
    ++rays[0];
    uint32_t inst = kInvalidId, prim = kInvalidId;
    float    bu = 0.0f, bv = 0.0f;
    if (hit.gid != kInvalidId)
    {
        inst = sc.tris[hit.gid].inst, prim = sc.tris[hit.gid].prim, bu = hit.u, bv = hit.v;
    }
    o->geo[0] = bu, o->geo[1] = bv, o->geo[2] = as_float(inst), o->geo[3] = as_float(prim);

    LightSample ls = directional_light(frame_count);
    auto set4 = [](float* d, float a, float b, float c, float e) { d[0] = a, d[1] = b, d[2] = c, d[3] = e; };

    // ---- rt_direct_lighting.hlsl:38-83 ----
    if (inst == kInvalidId)
    {
        set4(o->direct, 0.7f, 0.7f, 0.85f, 1.0f);
        set4(o->albedo, 1, 1, 1, 1);
        set4(o->nd, 0, 0, 0, 0);
        set4(o->indirect, 0, 0, 0, 1);  // rt_indirect.hlsl:75-79
        return;
    }
    f3    p, n;
    float tx[2];
    interpolate_attributes(sc, inst, prim, bu, bv, &p, &n, tx);
    f3 kd = get_material(sc, inst, tx);
    if (kd.x < 1e-5f && kd.y < 1e-5f && kd.z < 1e-5f)
    {
        set4(o->direct, 0, 0, 0, 1);
        set4(o->albedo, 0, 0, 0, 0);
        set4(o->nd, 0, 0, 0, 0);
    }
    else
    {
        f3 di = direct_illumination(sc, ls, p, n, kd, use_bvh, &rays[2]);
        set4(o->direct, di.x, di.y, di.z, 1.0f);
        set4(o->albedo, kd.x, kd.y, kd.z, 1.0f);
        float oct[2];
        oct_encode(n, oct);
        f3 cp = make3(cam.position[0], cam.position[1], cam.position[2]) - p;
        set4(o->nd, oct[0], oct[1], (float)inst, length(cp));
    }

    // ---- rt_indirect.hlsl:82-176 ----
    if (lowres)
    {
        // LOWRES_INDIRECT (:53-59): the pass runs on the half-resolution grid; of every 2x2 full-resolution block only the pixel
        // at sp_offset = ((frame % 4) / 2, (frame % 4) % 2) gets an indirect sample this frame
        const uint32_t ox = (frame_count % 4) / 2, oy = (frame_count % 4) % 2;
        if ((x & 1u) != ox || (y & 1u) != oy)
        {
            set4(o->indirect, 0, 0, 0, 0);
            return;
        }
    }
    f3 color = make3(0, 0, 0), thr = make3(1, 1, 1);
    for (uint32_t bounce = 0; bounce <= num_bounces; ++bounce)
    {
        if (inst == kInvalidId)
        {
            color = color + thr * kSky;  // :97
            break;
        }
        if (bounce != 0)  // bounce 0 attributes were fetched above (same values as :103-105)
        {
            interpolate_attributes(sc, inst, prim, bu, bv, &p, &n, tx);
            kd = get_material(sc, inst, tx);
        }
        if (kd.x < 1e-5f && kd.y < 1e-5f && kd.z < 1e-5f) break;  // :108
        if (bounce != 0 && fb)
        {
            // :116-145 GBUFFER_FEEDBACK: a vertex the previous frame saw returns that frame's shaded result and ends the path.
            // Stated choice: a NaN uv counts as disocclusion (any(uv < 0) || any(uv > 1) is false for NaN in HLSL and would
            // reach an undefined texel address).
            f2   puv          = image_plane_uv(fb->prev_cam, p);
            bool disocclusion = !(puv.x >= 0.0f && puv.y >= 0.0f && puv.x <= 1.0f && puv.y <= 1.0f);
            if (!disocclusion)
            {
                f2    pxy = uv_to_xy(puv, w, h);
                int   ix = (int)pxy.x, iy = (int)pxy.y;  // Load(int3(prev_frame_xy, 0))
                float prev_depth = (ix >= 0 && iy >= 0 && (uint32_t)ix < w && (uint32_t)iy < h)
                                       ? fb->prev_normal_depth[4 * ((size_t)iy * w + ix) + 3]
                                       : 0.0f;
                float cur_depth = length(p - make3(fb->prev_cam.position[0], fb->prev_cam.position[1], fb->prev_cam.position[2]));
                disocclusion    = fabsf(prev_depth - cur_depth) / cur_depth > 0.05f;
            }
            if (!disocclusion)
            {
                color = color + thr * sample_bilinear(fb->color_history, w, h, puv);  // :142
                break;
            }
        }
        if (bounce != 0) color = color + thr * direct_illumination(sc, ls, p, n, kd, use_bvh, &rays[2]);  // :136
        float s[2];
        bluenoise4x4(bn, x, y, frame_count * 25 + bounce, s);  // :149
        f3    d   = map_to_hemisphere(s, n);                   // shading.h:24-32
        float ndd = dot(n, d);
        float pdf = hmax(0.0f, ndd) / kPi;                     // shading.h:19-22
        if (pdf < 1e-5f) break;                                // :160
        float f = (kInvPi * hmax(ndd, 0.0f)) / pdf;            // :165
        thr     = thr * f;
        if (bounce != 0) thr = thr * kd;                       // :167-170
        if (bounce == num_bounces) break;  // the reference traces one more ray whose payload is never read (:91,:173)
        hit = trace_closest(sc, p, d, 0.0001f, 100000.0f, use_bvh);  // :154-157,:173
        ++rays[1];
        if (hit.gid != kInvalidId)
            inst = sc.tris[hit.gid].inst, prim = sc.tris[hit.gid].prim, bu = hit.u, bv = hit.v;
        else
            inst = prim = kInvalidId;  // Miss, :187-192
    }
    set4(o->indirect, color.x, color.y, color.z, 1.0f);
}

// ---------------------------------------------------------------------------------------------
// EXT shading model (GGX, emissive triangles, next-event estimation).  No reference counterpart:
// the specification is this code (DESIGN.md "EXT shading model"); the HIP path implements it independently.
//   BSDF  f = kd/pi + ks * D_ggx(h) * G_smith(wo, wi) / (4 (n.wo)(n.wi)),  alpha = max(roughness^2, 1e-3), shading normal
//         flipped towards wo;  lobe choice by luminance;  pdf = ps * D (n.h) / (4 wo.h) + (1 - ps) * (n.wi)/pi
//   light emissive triangles sampled uniformly by area (float prefix sums in triangle order), one shadow ray per vertex;
//         emission is seen directly only from the camera (bounce 0), every other vertex gets it through NEE
//   rng   blue-noise texel of (pixel, frame*25+bounce): R,G -> BSDF direction, B -> lobe, A -> light triangle;
//         texel of count+7: R,G -> point on the light.  Sky and the reference's directional light are off.
// ---------------------------------------------------------------------------------------------
void bluenoise4x4_rgba(const uint8_t* tex, uint32_t x, uint32_t y, uint32_t count, float out[4])
{
    uint32_t       px = (count % 16) % 4, py = (count % 16) / 4;
    uint32_t       sx = (x * 4 + px) % 256, sy = (y * 4 + py) % 256;
    const uint8_t* t  = tex + 4 * (sy * 256 + sx);
    float          k  = 0.61803398875f * (float)(count / 16);
    for (int c = 0; c < 4; ++c) out[c] = frac((float)t[c] / 255.0f + k);
}

inline float lum(f3 c) { return fmaf(c.z, 0.114f, fmaf(c.y, 0.587f, c.x * 0.299f)); }  // math_functions.h:25-28 weights

struct ExtBsdf
{
    f3    f;
    float pdf_spec, pdf_diff;
};
ExtBsdf ext_bsdf(f3 kd, f3 ks, float a2, f3 nf, f3 wo, f3 wi)
{
    float cos_o = dot(nf, wo), cos_i = dot(nf, wi);
    f3    h     = normalize(wo + wi);
    float cos_h = dot(nf, h), woh = dot(wo, h);
    float dd    = fmaf(cos_h * cos_h, a2 - 1.0f, 1.0f);
    // D G / (4 cos_o cos_i) with Smith's G = g_o g_i, g = 2 cos / (cos + sqrt(a2 + (1 - a2) cos^2)), in its cancelled form: the
    // 2 cos of either g and the 4 cos_o cos_i go, leaving ONE division, D V = a2 / (pi dd^2 (cos_o + s_o)(cos_i + s_i)) -- the
    // "visibility" form of the same function (EXT shading model; until round 6 the five divisions were written out)
    float pdd   = kPi * dd * dd;
    float lam_o = cos_o + sqrtf(fmaf(1.0f - a2, cos_o * cos_o, a2));
    float lam_i = cos_i + sqrtf(fmaf(1.0f - a2, cos_i * cos_i, a2));
    float spec  = a2 / (pdd * (lam_o * lam_i));
    ExtBsdf r;
    r.f        = make3(kd.x * kInvPi + ks.x * spec, kd.y * kInvPi + ks.y * spec, kd.z * kInvPi + ks.z * spec);
    r.pdf_spec = (a2 * cos_h) / (pdd * (4.0f * woh));
    r.pdf_diff = cos_i * kInvPi;
    return r;
}

void tri_positions(const Scene& sc, uint32_t gid, f3* p0, f3* p1, f3* p2)
{
    const Tri&        tr   = sc.tris[gid];
    const OracleMesh& mesh = sc.meshes[tr.inst];
    const uint32_t*   ix   = &sc.indices[mesh.first_index_offset + 3 * tr.prim];
    auto              P    = [&](uint32_t i) {
        uint32_t v = mesh.first_vertex_offset + i;
        return make3(sc.positions[3 * v], sc.positions[3 * v + 1], sc.positions[3 * v + 2]);
    };
    *p0 = P(ix[0]), *p1 = P(ix[1]), *p2 = P(ix[2]);
}

void shade_pixel_ext(const Scene& sc, const OracleCamera& cam, const uint8_t* bn, uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                     uint32_t frame_count, uint32_t num_bounces, bool use_bvh, PixelOut* o, uint64_t rays[3])
{
    f3 org, dir;
    create_primary_ray(cam, x, y, w, h, frame_count, &org, &dir);
    Hit hit = trace_closest(sc, org, dir, 0.0f, 1e6f, use_bvh);
    ++rays[0];
    auto set4 = [](float* d, float a, float b, float c, float e) { d[0] = a, d[1] = b, d[2] = c, d[3] = e; };
    uint32_t inst = kInvalidId, prim = kInvalidId;
    float    bu = 0.0f, bv = 0.0f;
    if (hit.gid != kInvalidId) inst = sc.tris[hit.gid].inst, prim = sc.tris[hit.gid].prim, bu = hit.u, bv = hit.v;
    o->geo[0] = bu, o->geo[1] = bv, o->geo[2] = as_float(inst), o->geo[3] = as_float(prim);
    set4(o->albedo, 1, 1, 1, 1);
    set4(o->nd, 0, 0, 0, 0);
    f3 direct = make3(0, 0, 0), color = make3(0, 0, 0), thr = make3(1, 1, 1);
    for (uint32_t bounce = 0; bounce <= num_bounces; ++bounce)
    {
        if (inst == kInvalidId) break;  // environment is black in the EXT model
        f3    p, n;
        float tx[2];
        interpolate_attributes(sc, inst, prim, bu, bv, &p, &n, tx);
        const OracleMaterial& m = sc.materials[inst];
        f3    kd = make3(m.kd[0], m.kd[1], m.kd[2]), ks = make3(m.ks[0], m.ks[1], m.ks[2]), ke = make3(m.ke[0], m.ke[1], m.ke[2]);
        float alpha = hmax(m.roughness * m.roughness, 1e-3f), a2 = alpha * alpha;
        f3    wo = make3(-dir.x, -dir.y, -dir.z);
        f3    nf = dot(n, wo) < 0.0f ? make3(-n.x, -n.y, -n.z) : n;
        if (bounce == 0)
        {
            direct = ke;
            float oct[2];
            oct_encode(n, oct);
            f3 cp = make3(cam.position[0], cam.position[1], cam.position[2]) - p;
            set4(o->nd, oct[0], oct[1], (float)inst, length(cp));
        }
        float ra[4], rb[4];
        bluenoise4x4_rgba(bn, x, y, frame_count * 25 + bounce, ra);
        bluenoise4x4_rgba(bn, x, y, frame_count * 25 + bounce + 7, rb);
        // ---- next-event estimation ----
        if (!sc.light_tris.empty())
        {
            float  target = ra[3] * sc.light_area;
            size_t j      = 0;
            while (j + 1 < sc.light_tris.size() && !(sc.light_cdf[j] > target)) ++j;
            uint32_t lg = sc.light_tris[j];
            f3       q0, q1, q2;
            tri_positions(sc, lg, &q0, &q1, &q2);
            float su = sqrtf(rb[0]), b0 = 1.0f - su, b1 = su * (1.0f - rb[1]), b2 = su * rb[1];
            f3    pl = make3(fmaf(q2.x, b2, fmaf(q1.x, b1, q0.x * b0)), fmaf(q2.y, b2, fmaf(q1.y, b1, q0.y * b0)),
                             fmaf(q2.z, b2, fmaf(q1.z, b1, q0.z * b0)));
            f3    nl = normalize(cross(q1 - q0, q2 - q0));
            f3    Lv = pl - p;
            float d2 = dot(Lv, Lv), dist = sqrtf(d2);
            f3    wi = Lv * (1.0f / dist);
            float cos_s = dot(nf, wi), cos_l = fabsf(dot(nl, wi));
            if (cos_s > 0.0f && cos_l > 0.0f && d2 > 0.0f)
            {
                const OracleMaterial& lm = sc.materials[sc.tris[lg].inst];
                ExtBsdf bs = ext_bsdf(kd, ks, a2, nf, wo, wi);
                float   wgt = ((cos_s * cos_l) * sc.light_area) / d2;
                f3      c = make3((thr.x * bs.f.x) * (lm.ke[0] * wgt), (thr.y * bs.f.y) * (lm.ke[1] * wgt), (thr.z * bs.f.z) * (lm.ke[2] * wgt));
                if (c.x != 0.0f || c.y != 0.0f || c.z != 0.0f)
                {
                    ++rays[2];
                    if (!trace_any(sc, p, wi, 0.0001f, dist * 0.999f, use_bvh))
                    {
                        if (bounce == 0) direct = direct + c; else color = color + c;
                    }
                }
            }
        }
        // ---- BSDF sampling ----
        float ls = lum(ks), sum = lum(kd) + ls;
        if (!(sum > 0.0f)) break;
        float ps = ls / sum;
        f3    wi;
        if (ra[2] < ps)
        {
            float c2 = (1.0f - ra[1]) / fmaf(a2 - 1.0f, ra[1], 1.0f);
            float ct = sqrtf(c2), st = sqrtf(hmax(0.0f, 1.0f - c2));
            float sp, cp;
            sincos_contract((2.0f * kPi) * ra[0], &sp, &cp);
            f3 uu = ortho_vector(nf);
            f3 vv = cross(uu, nf);
            uu    = cross(nf, vv);
            float a = st * cp, b = st * sp;
            f3 hh = normalize(make3(fmaf(nf.x, ct, fmaf(vv.x, b, uu.x * a)), fmaf(nf.y, ct, fmaf(vv.y, b, uu.y * a)),
                                    fmaf(nf.z, ct, fmaf(vv.z, b, uu.z * a))));
            float k2 = 2.0f * dot(wo, hh);
            wi = make3(fmaf(hh.x, k2, -wo.x), fmaf(hh.y, k2, -wo.y), fmaf(hh.z, k2, -wo.z));
        }
        else
        {
            float s2[2] = {ra[0], ra[1]};
            wi = map_to_hemisphere(s2, nf);
        }
        float cos_i = dot(nf, wi);
        if (!(cos_i > 0.0f)) break;
        ExtBsdf bs  = ext_bsdf(kd, ks, a2, nf, wo, wi);
        float   pdf = ps * bs.pdf_spec + (1.0f - ps) * bs.pdf_diff;
        if (!(pdf > 1e-8f)) break;
        float wgt = cos_i / pdf;
        thr = make3(thr.x * (bs.f.x * wgt), thr.y * (bs.f.y * wgt), thr.z * (bs.f.z * wgt));
        if (bounce == num_bounces) break;
        dir = wi;
        hit = trace_closest(sc, p, wi, 0.0001f, 100000.0f, use_bvh);
        ++rays[1];
        if (hit.gid != kInvalidId)
            inst = sc.tris[hit.gid].inst, prim = sc.tris[hit.gid].prim, bu = hit.u, bv = hit.v;
        else
            inst = prim = kInvalidId;
    }
    set4(o->direct, direct.x, direct.y, direct.z, 1.0f);
    set4(o->indirect, color.x, color.y, color.z, 1.0f);
}

void render_rows(const Scene& sc, const OracleCamera& cam, const uint8_t* bn, uint32_t w, uint32_t h, uint32_t frame,
                 uint32_t bounces, bool use_bvh, bool ext, uint32_t row0, uint32_t row_step, OracleFrameOutputs* out, uint64_t rays[3],
                 const Feedback* fb, bool lowres, uint32_t row_end)
{
    for (uint32_t y = row0; y < row_end; y += row_step)
        for (uint32_t x = 0; x < w; ++x)
        {
            PixelOut po;
            if (ext)
                shade_pixel_ext(sc, cam, bn, x, y, w, h, frame, bounces, use_bvh, &po, rays);
            else
                shade_pixel(sc, cam, bn, x, y, w, h, frame, bounces, use_bvh, &po, rays, fb, lowres);
            size_t i = 4 * ((size_t)y * w + x);
            if (out->gbuffer_geo) memcpy(out->gbuffer_geo + i, po.geo, 16);
            if (out->direct) memcpy(out->direct + i, po.direct, 16);
            if (out->albedo) memcpy(out->albedo + i, po.albedo, 16);
            if (out->normal_depth) memcpy(out->normal_depth + i, po.nd, 16);
            if (out->indirect) memcpy(out->indirect + i, po.indirect, 16);
            if (lowres && out->indirect_lowres && (x & 1u) == (frame % 4) / 2 && (y & 1u) == (frame % 4) % 2)
                memcpy(out->indirect_lowres + 4 * ((size_t)(y >> 1) * (w >> 1) + (x >> 1)), po.indirect, 16);  // g_output_indirect[xy]
            if (out->combined)
                for (int c = 0; c < 4; ++c)  // combine_illumination.hlsl:24,29 (indirect.w forced to 1)
                    out->combined[i + c] = (c == 3 ? 1.0f : po.indirect[c]) * po.albedo[c] + po.direct[c];
        }
}
}  // namespace

extern "C" {

void* oracle_scene_create(const OracleScene* s)
{
    Scene* sc = new Scene;
    sc->positions.assign(s->positions, s->positions + 3 * (size_t)s->vertex_count);
    sc->normals.assign(s->normals, s->normals + 3 * (size_t)s->vertex_count);
    sc->texcoords.assign(s->texcoords, s->texcoords + 2 * (size_t)s->vertex_count);
    sc->indices.assign(s->indices, s->indices + s->index_count);
    sc->meshes.assign(s->meshes, s->meshes + s->mesh_count);
    for (uint32_t i = 0; i < s->texture_count; ++i)
    {
        const OracleTexture& t = s->textures[i];
        sc->texture_data.emplace_back(t.rgba8, t.rgba8 + 4 * (size_t)t.width * t.height);
    }
    for (uint32_t i = 0; i < s->texture_count; ++i)
        sc->textures.push_back(OracleTexture{sc->texture_data[i].data(), s->textures[i].width, s->textures[i].height});
    if (s->materials) sc->materials.assign(s->materials, s->materials + s->mesh_count);
    for (uint32_t m = 0; m < s->mesh_count; ++m)
    {
        const OracleMesh& mesh = sc->meshes[m];
        for (uint32_t k = 0; k + 2 < mesh.index_count; k += 3)
        {
            f3 v[3];
            for (int j = 0; j < 3; ++j)
            {
                uint32_t vi = mesh.first_vertex_offset + sc->indices[mesh.first_index_offset + k + j];
                v[j]        = make3(sc->positions[3 * vi], sc->positions[3 * vi + 1], sc->positions[3 * vi + 2]);
            }
            f3 e1 = v[1] - v[0], e2 = v[2] - v[0];
            sc->tris.push_back(Tri{v[0], e1, e2, cross(e1, e2), mesh.index, k / 3});
        }
    }
    if (!sc->materials.empty())
        for (uint32_t g = 0; g < sc->tris.size(); ++g)
        {
            const OracleMaterial& m = sc->materials[sc->tris[g].inst];
            if (m.ke[0] > 0.0f || m.ke[1] > 0.0f || m.ke[2] > 0.0f)
            {
                sc->light_area = sc->light_area + 0.5f * length(sc->tris[g].n);
                sc->light_tris.push_back(g);
                sc->light_cdf.push_back(sc->light_area);
            }
        }
    build_bvh(*sc);
    return sc;
}

void oracle_scene_destroy(void* h) { delete (Scene*)h; }

// rows [y0, y1) of the frame are rendered (the whole frame: 0, h); pixels outside keep what the output buffers held
static int render_frame_impl(void* scene, const OracleCamera* cam, const uint8_t* bn, uint32_t w, uint32_t h, uint32_t frame_count,
                             uint32_t num_bounces, uint32_t flags, uint32_t num_threads, OracleFrameOutputs* out, const Feedback* fb,
                             uint32_t y0 = 0, uint32_t y1 = ~0u)
{
    if (!scene || !cam || !bn || !out || !w || !h) return 1;
    y1 = std::min(y1, h);
    if (y0 >= y1) return 1;
    const Scene& sc = *(const Scene*)scene;
    bool     bvh = (flags & ORACLE_FLAG_USE_BVH) != 0;
    bool     ext = (flags & ORACLE_FLAG_EXT_MATERIALS) != 0;
    if (ext && sc.materials.size() != sc.meshes.size()) return 2;  // EXT needs one material per mesh
    if (ext && fb) return 3;                                        // the feedback branch belongs to the reference model
    const bool lowres = (flags & ORACLE_FLAG_LOWRES_INDIRECT) != 0;
    if (lowres && (ext || (w & 1u) || (h & 1u))) return 4;          // half-resolution indirect: reference model, even extents
    uint32_t nt  = std::max(1u, std::min(num_threads, y1 - y0));
    std::vector<uint64_t> rays(3 * (size_t)nt, 0);
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < nt; ++t)
        th.emplace_back(render_rows, std::cref(sc), std::cref(*cam), bn, w, h, frame_count, num_bounces, bvh, ext, y0 + t, nt, out,
                        rays.data() + 3 * t, fb, lowres, y1);
    render_rows(sc, *cam, bn, w, h, frame_count, num_bounces, bvh, ext, y0, nt, out, rays.data(), fb, lowres, y1);
    for (auto& t : th) t.join();
    out->rays[0] = out->rays[1] = out->rays[2] = 0;
    for (uint32_t t = 0; t < nt; ++t)
        for (int k = 0; k < 3; ++k) out->rays[k] += rays[3 * t + k];
    return 0;
}

int oracle_render_frame(void* scene, const OracleCamera* cam, const uint8_t* bn, uint32_t w, uint32_t h, uint32_t frame_count,
                        uint32_t num_bounces, uint32_t flags, uint32_t num_threads, OracleFrameOutputs* out)
{
    return render_frame_impl(scene, cam, bn, w, h, frame_count, num_bounces, flags, num_threads, out, nullptr);
}

// Rows [y0, y1) only: oracle crops of frames too large to restate whole in a test (BASELINE configs 3 and 5).
int oracle_render_frame_rows(void* scene, const OracleCamera* cam, const uint8_t* bn, uint32_t w, uint32_t h, uint32_t frame_count,
                             uint32_t num_bounces, uint32_t flags, uint32_t num_threads, uint32_t y0, uint32_t y1, OracleFrameOutputs* out)
{
    return render_frame_impl(scene, cam, bn, w, h, frame_count, num_bounces, flags, num_threads, out, nullptr, y0, y1);
}

int oracle_render_frame_feedback(void* scene, const OracleCamera* cam, const OracleCamera* prev_cam, const uint8_t* bn, uint32_t w,
                                 uint32_t h, uint32_t frame_count, uint32_t num_bounces, uint32_t flags, uint32_t num_threads,
                                 const float* prev_normal_depth, const float* color_history, OracleFrameOutputs* out)
{
    if (!prev_cam || !prev_normal_depth || !color_history) return 1;
    Feedback fb{*prev_cam, prev_normal_depth, color_history};
    return render_frame_impl(scene, cam, bn, w, h, frame_count, num_bounces, flags, num_threads, out, &fb);
}

int oracle_render_accumulate(void* scene, const OracleCamera* cam, const uint8_t* bn, uint32_t w, uint32_t h,
                             uint32_t frame_begin, uint32_t n_frames, uint32_t num_bounces, uint32_t flags,
                             uint32_t num_threads, float* accum, uint64_t rays[3])
{
    std::vector<float> combined(4 * (size_t)w * h);
    OracleFrameOutputs out;
    memset(&out, 0, sizeof(out));
    out.combined = combined.data();
    if (rays) rays[0] = rays[1] = rays[2] = 0;
    for (uint32_t f = 0; f < n_frames; ++f)
    {
        int rc = oracle_render_frame(scene, cam, bn, w, h, frame_begin + f, num_bounces, flags, num_threads, &out);
        if (rc) return rc;
        for (size_t i = 0; i < combined.size(); ++i) accum[i] = accum[i] + combined[i];
        if (rays)
            for (int k = 0; k < 3; ++k) rays[k] += out.rays[k];
    }
    return 0;
}

void     oracle_halton23(uint32_t fc, float out[2]) { halton23(fc, out); }
uint32_t oracle_wang_hash(uint32_t x, uint32_t y) { return wang_hash(x, y); }
void     oracle_bluenoise4x4(const uint8_t* t, uint32_t x, uint32_t y, uint32_t c, float out[2]) { bluenoise4x4(t, x, y, c, out); }
void     oracle_directional_light(uint32_t count, float dir[3], float inten[3])
{
    LightSample ls = directional_light(count);
    dir[0] = ls.direction.x, dir[1] = ls.direction.y, dir[2] = ls.direction.z;
    inten[0] = ls.intensity.x, inten[1] = ls.intensity.y, inten[2] = ls.intensity.z;
}
void oracle_primary_ray(const OracleCamera* cam, uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t fc, float o[3],
                        float d[3])
{
    f3 oo, dd;
    create_primary_ray(*cam, x, y, w, h, fc, &oo, &dd);
    o[0] = oo.x, o[1] = oo.y, o[2] = oo.z, d[0] = dd.x, d[1] = dd.y, d[2] = dd.z;
}
void oracle_map_to_hemisphere(const float s[2], const float n[3], float out[3])
{
    f3 d = map_to_hemisphere(s, make3(n[0], n[1], n[2]));
    out[0] = d.x, out[1] = d.y, out[2] = d.z;
}
void  oracle_sincos(float x, float* s, float* c) { sincos_contract(x, s, c); }
float oracle_pow22(float x) { return pow22_contract(x); }
void  oracle_oct_encode(const float n[3], float out[2]) { oct_encode(make3(n[0], n[1], n[2]), out); }
int   oracle_intersect_triangle(const float o[3], const float d[3], float tmin, float tmax, const float v0[3],
                                const float v1[3], const float v2[3], float* t, float* u, float* v)
{
    f3  a = make3(v0[0], v0[1], v0[2]), b = make3(v1[0], v1[1], v1[2]), c = make3(v2[0], v2[1], v2[2]);
    Tri tr{a, b - a, c - a, cross(b - a, c - a), 0, 0};
    return intersect_tri(make3(o[0], o[1], o[2]), make3(d[0], d[1], d[2]), tr, tmin, tmax, t, u, v) ? 1 : 0;
}
void oracle_sample_texture(const OracleTexture* tex, float u, float v, float out[3]) { sample_texture(*tex, u, v, out); }
}
