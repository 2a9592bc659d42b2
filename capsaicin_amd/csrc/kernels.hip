// kernels.hip — gfx950 (CDNA4) kernels of the wavefront path tracer: BVH traversal / triangle intersection,
// shading + BSDF sampling + 64-lane queue compaction, radiance accumulate and tile exchange helpers.
//
// Reference semantics (paths relative to /root/reference/src/core/shaders): rt_primary_visibility.hlsl,
// rt_direct_lighting.hlsl, rt_indirect.hlsl, camera.h, sampling.h, lighting.h, shading.h, scene.h.
// The DXR TraceRay / acceleration structure (driver code in the reference) is replaced by the explicit
// traversal below.  No MFMA: nothing here is a dense contraction.
#include "cap_kernels.h"
#include "cap_reproject.h"
#include "cap_trace.h"
#include "cap_wide_trace.h"

namespace cap
{
// Closest hit: minimum t, equal t resolved towards the lower global triangle id (visit-order independent).
// stack: this lane's column of the per-wave LDS stack; entry k lives at stack[k * kBlock].
template <int STACK>
__device__ __forceinline__ void traverse_closest(const BvhDev& bvh, const Ray& r, uint32_t* stack, float& best_t, float& best_u,
                                                 float& best_v, uint32_t& best_gid)
{
    best_t = r.tmax, best_u = 0.0f, best_v = 0.0f, best_gid = kInvalidId;
    if (bvh.tri_count == 0) return;
    int node = bvh.root;
    int sp   = 0;
    while (true)
    {
        if (node >= 0)
        {
            const float4 q0 = bvh.nodes[4 * node + 0], q1 = bvh.nodes[4 * node + 1], q2 = bvh.nodes[4 * node + 2],
                         q3 = bvh.nodes[4 * node + 3];
            float      tn0, tn1;
            const bool h0 = slab(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, best_t, tn0);
            const bool h1 = slab(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, best_t, tn1);
            const int  c0 = (int)f2u(q3.z), c1 = (int)f2u(q3.w);
            if (h0 && h1)
            {
                const bool swap = tn1 < tn0;
                if (sp < STACK) stack[(sp++) * kBlock] = (uint32_t)(swap ? c0 : c1);
                node = swap ? c1 : c0;
                continue;
            }
            if (h0 || h1)
            {
                node = h0 ? c0 : c1;
                continue;
            }
        }
        else
        {
            // leaf = up to kLeafMax consecutive sorted triangles: ~(first | (count - 1) << kLeafCountShift)
            const uint32_t code = (uint32_t)~node, first = code & kLeafFirstMask, last = first + (code >> kLeafCountShift);
            for (uint32_t leaf = first; leaf <= last; ++leaf)
            {
                const float4 t0 = bvh.tris[4 * leaf + 0], t1 = bvh.tris[4 * leaf + 1], t2 = bvh.tris[4 * leaf + 2];
                float        t, u, v;
                if (tri_test(r, t0, t1, t2, t, u, v))
                {
                    const uint32_t gid = f2u(bvh.tris[4 * leaf + 3].x);
                    if (t < best_t || (t == best_t && gid < best_gid)) best_t = t, best_u = u, best_v = v, best_gid = gid;
                }
            }
        }
        if (sp == 0) break;
        node = (int)stack[(--sp) * kBlock];
    }
}

// Any hit (RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH, lighting.h:49): true when some triangle has tmin < t < tmax.
template <int STACK>
__device__ __forceinline__ bool traverse_any(const BvhDev& bvh, const Ray& r, uint32_t* stack)
{
    if (bvh.tri_count == 0) return false;
    if (bvh.wide8_ok) return traverse_any8<STACK>(bvh, r, stack);  // wave-uniform
    int node = bvh.root;
    int sp   = 0;
    while (true)
    {
        if (node >= 0)
        {
            const float4 q0 = bvh.nodes[4 * node + 0], q1 = bvh.nodes[4 * node + 1], q2 = bvh.nodes[4 * node + 2],
                         q3 = bvh.nodes[4 * node + 3];
            float      tn0, tn1;
            const bool h0 = slab(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, r.tmax, tn0);
            const bool h1 = slab(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, r.tmax, tn1);
            const int  c0 = (int)f2u(q3.z), c1 = (int)f2u(q3.w);
            if (h0 && h1)
            {
                if (sp < STACK) stack[(sp++) * kBlock] = (uint32_t)c1;
                node = c0;
                continue;
            }
            if (h0 || h1)
            {
                node = h0 ? c0 : c1;
                continue;
            }
        }
        else
        {
            const uint32_t code = (uint32_t)~node, first = code & kLeafFirstMask, last = first + (code >> kLeafCountShift);
            for (uint32_t leaf = first; leaf <= last; ++leaf)
            {
                const float4 t0 = bvh.tris[4 * leaf + 0], t1 = bvh.tris[4 * leaf + 1], t2 = bvh.tris[4 * leaf + 2];
                if (tri_occludes(r, t0, t1, t2)) return true;
            }
        }
        if (sp == 0) break;
        node = (int)stack[(--sp) * kBlock];
    }
    return false;
}

// Small scenes (tri_count <= kExhaustiveMax): the hierarchy degenerates to one leaf holding every triangle, tested
// exhaustively.  The loop counter is wave-uniform, so the triangle records are fetched once per wave through the scalar
// data cache (s_load) instead of 64 times through the vector path, and no lane ever waits for another lane's traversal:
// 64-lane SIMD efficiency is 100 % whatever the ray distribution.  Same hit rule, so the same answer as the stack traversal.
// The triangle array is written once by the BVH build and never during a render, so it may be read through the constant
// address space: with a wave-uniform index the compiler then emits s_load (scalar data cache, operands in SGPRs).
struct RawF4
{
    float x, y, z, w;
};
struct alignas(64) RawTri
{
    RawF4 q[4];
};
// One 64-byte triangle record per s_load_dwordx16.
__device__ __forceinline__ void load_const_tri(const float4* base, uint32_t k, float4& t0, float4& t1, float4& t2, float4& t3)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const RawTri ConstTri;
    const RawTri v = ((const ConstTri*)base)[k];
    t0 = make_float4(v.q[0].x, v.q[0].y, v.q[0].z, v.q[0].w);
    t1 = make_float4(v.q[1].x, v.q[1].y, v.q[1].z, v.q[1].w);
    t2 = make_float4(v.q[2].x, v.q[2].y, v.q[2].z, v.q[2].w);
    t3 = make_float4(v.q[3].x, v.q[3].y, v.q[3].z, v.q[3].w);
#else
    t0 = base[4 * k], t1 = base[4 * k + 1], t2 = base[4 * k + 2], t3 = base[4 * k + 3];
#endif
}

// The determinant-scaled quantities of tri_test() for one triangle record, sign-flipped so that det >= 0.  Same values bit for
// bit: det = -(d.n) and V = -(e1.q) are exact negations, so their sign bits are folded into the flip masks instead of being
// applied first (three bit operations instead of five).
struct TriScaled
{
    float det, U, V, T;  // det = |d.n|
};
__device__ __forceinline__ TriScaled tri_scaled(const Ray& r, const float4 t0, const float4 t1, const float4 t2)
{
    const v3 v0 = mk3(t0.x, t0.y, t0.z), e1 = mk3(t0.w, t1.x, t1.y), e2 = mk3(t1.z, t1.w, t2.x), n = mk3(t2.y, t2.z, t2.w);
    const v3 tvec = r.o - v0;
    const v3 q    = cross3(tvec, r.d);
    const float    ddn = dot3(r.d, n);                 // det = -ddn
    const uint32_t s   = f2u(ddn) & 0x80000000u;       // sign of ddn = NOT sign of det
    TriScaled o;
    o.det = fabsf(ddn);
    o.U   = u2f(f2u(dot3(e2, q)) ^ (s ^ 0x80000000u));  // U ^ sign(det)
    o.V   = u2f(f2u(dot3(e1, q)) ^ s);                  // (-e1.q) ^ sign(det)
    o.T   = u2f(f2u(dot3(tvec, n)) ^ (s ^ 0x80000000u));
    return o;
}

// Fan pair: triangles id and id + 1 share v0 and the edge e2(id) == e1(id + 1) (every triangulated quad), so tvec, q and that
// edge's dot product with q are computed once.  20 floats: (v0, e1, e2, e3, nA, nB, asfloat(id), 0), read through the scalar
// cache like the single records.  The per-triangle arithmetic is exactly tri_scaled()'s.
struct alignas(16) RawPair
{
    float f[20];
};
struct PairScaled
{
    TriScaled a, b;
    uint32_t  id;
};
// ORG: every ray of the launch has the same origin (camera rays), so tvec = o - v0 and T = tvec.n of both triangles are the same
// for every ray; org_tab holds them per pair -- (tvec, tvec.nA) (tvec.nB, -, -, -), computed once per workgroup with the operations
// below -- and the loop reads them from LDS (one address for the whole wave) instead of spending nine vector instructions.
template <bool ORG = false>
__device__ __forceinline__ PairScaled pair_scaled(const Ray& r, const float4* base, uint32_t k, const float4* org_tab = nullptr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const RawPair ConstPair;
    const RawPair p = ((const ConstPair*)base)[k];
#else
    RawPair p;
    for (int i = 0; i < 20; ++i) p.f[i] = reinterpret_cast<const float*>(base)[20 * k + i];
#endif
    const v3 v0 = mk3(p.f[0], p.f[1], p.f[2]), e1 = mk3(p.f[3], p.f[4], p.f[5]), e2 = mk3(p.f[6], p.f[7], p.f[8]),
             e3 = mk3(p.f[9], p.f[10], p.f[11]), na = mk3(p.f[12], p.f[13], p.f[14]), nb = mk3(p.f[15], p.f[16], p.f[17]);
    v3    tvec;
    float tna, tnb;
    if (ORG)
    {
        const float4 c0 = org_tab[2 * k], c1 = org_tab[2 * k + 1];
        tvec = mk3(c0.x, c0.y, c0.z), tna = c0.w, tnb = c1.x;
    }
    else
        tvec = r.o - v0, tna = dot3(tvec, na), tnb = dot3(tvec, nb);
    const v3 q    = cross3(tvec, r.d);
    const float e2q = dot3(e2, q);  // U of the first triangle, V (before its negation) of the second
    PairScaled  o;
    {
        const float    ddn = dot3(r.d, na);
        const uint32_t s   = f2u(ddn) & 0x80000000u;
        o.a.det = fabsf(ddn);
        o.a.U   = u2f(f2u(e2q) ^ (s ^ 0x80000000u));
        o.a.V   = u2f(f2u(dot3(e1, q)) ^ s);
        o.a.T   = u2f(f2u(tna) ^ (s ^ 0x80000000u));
    }
    {
        const float    ddn = dot3(r.d, nb);
        const uint32_t s   = f2u(ddn) & 0x80000000u;
        o.b.det = fabsf(ddn);
        o.b.U   = u2f(f2u(dot3(e3, q)) ^ (s ^ 0x80000000u));
        o.b.V   = u2f(f2u(e2q) ^ s);
        o.b.T   = u2f(f2u(tnb) ^ (s ^ 0x80000000u));
    }
    o.id = f2u(p.f[18]);
    return o;
}

// Exhaustive closest hit.  Same rule as tri_test() + "minimum t, ties to the lower id": both record lists are in ascending id
// order, so within a list "first strictly smaller t" is that rule (an equal t never replaces an earlier triangle); the two
// lists' winners are merged by the explicit (t, id) order.  With best_t starting at tmax, t < best_t implies t < tmax.  Only t
// and the id are tracked in the loops; the barycentrics of the winner are recomputed once afterwards (identical operations,
// identical bits) instead of being multiplied out and selected for every triangle.
// One candidate per triangle: its t, or +inf when the ray misses it.  Four triangles are tested side by side and reduced by
// a tree instead of a sequential compare chain (measured faster: more independent instructions for the scheduler to
// interleave); "(t, id) lexicographic minimum" is associative, so the tree gives the winner of the sequential rule.
// rec_tab: where the winner's record is re-read from (bvh.tris_by_id, or its LDS copy).
// MANY: the scene may hold more than 32 fan pairs (only in forced exhaustive mode beyond kExhaustiveMax triangles)
template <bool ORG = false, bool MANY = true>
__device__ __forceinline__ void exhaustive_closest(const BvhDev& bvh, const float4* rec_tab, const Ray& r, float& best_t, float& best_u,
                                                   float& best_v, uint32_t& best_gid, const float4* org_tab = nullptr,
                                                   uint32_t pair_mask = ~0u)
{
    best_t = r.tmax, best_u = 0.0f, best_v = 0.0f, best_gid = kInvalidId;
    auto cand = [&](const TriScaled& s) {
        const bool  inside = (s.U >= 0.0f) & (s.V >= 0.0f) & (s.U + s.V <= s.det);
        const float tt     = s.T * rcp_c(s.det);
        return (inside & (tt > r.tmin)) ? tt : __builtin_inff();
    };
    // ---- fan pairs, two at a time ----
    // A ray is inside at most one triangle of a planar quad except on the shared diagonal, so a pair needs one reciprocal: of the
    // triangle the ray is inside of.  Its t is bit for bit the one the per-triangle rule computes; the other triangle's
    // candidate is +inf either way.  Pairs some lane is inside both triangles of (non-planar fans, rays on the diagonal) take
    // the two-reciprocal form for the whole wave (wave-uniform branch, rare).
    auto pair_cand = [&](const PairScaled& p, float& m, uint32_t& im) {
        const bool a0 = p.a.U >= 0.0f, a1 = p.a.V >= 0.0f, a2 = p.a.U + p.a.V <= p.a.det;
        const bool b0 = p.b.U >= 0.0f, b1 = p.b.V >= 0.0f, b2 = p.b.U + p.b.V <= p.b.det;
        const bool ia = a0 & a1 & a2, ib = b0 & b1 & b2;
        // "some lane is inside both": the six compares' own lane masks AND-ed in scalar registers.  The ballot of the computed bool
        // (ia & ib) instead costs four half-rate vector instructions per pair -- two 0 / 1 materialisations, an AND and a compare -- and
        // keeps the 0 / 1 words alive for the compiler to build (ia | ib) and the id from: 20 issue cycles of a pair's ~165 (round 6, (87)).
        const unsigned long long both = __builtin_amdgcn_ballot_w64(a0) & __builtin_amdgcn_ballot_w64(a1) & __builtin_amdgcn_ballot_w64(a2) &
                                        __builtin_amdgcn_ballot_w64(b0) & __builtin_amdgcn_ballot_w64(b1) & __builtin_amdgcn_ballot_w64(b2);
        if (__builtin_expect(both != 0ull, 0))
        {
            const float ta = p.a.T * rcp_c(p.a.det), tb = p.b.T * rcp_c(p.b.det);
            const float ca = (ia & (ta > r.tmin)) ? ta : __builtin_inff(), cb = (ib & (tb > r.tmin)) ? tb : __builtin_inff();
            const bool  pb = cb < ca;  // strict: the earlier triangle keeps an equal t
            m = pb ? cb : ca, im = pb ? p.id + 1u : p.id;
        }
        else
        {
            const float tt = (ib ? p.b.T : p.a.T) * rcp_c(ib ? p.b.det : p.a.det);
            m  = ((ia | ib) & (tt > r.tmin)) ? tt : __builtin_inff();
            im = ib ? p.id + 1u : p.id;
        }
    };
    // pair_mask (wave-uniform): the fan pairs some ray of the wave can reach at all (camera rays: the pairs whose screen bounds
    // overlap the wave's tile, k_trace_shade); visited in ascending order, like the full list
    const uint32_t np = bvh.fan_pair_count;
    uint32_t       pm = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pair_mask & (np >= 32u ? ~0u : ((1u << np) - 1u))));
    if (MANY && np > 32u)
    {
        // more pairs than the mask holds (forced exhaustive mode on a larger scene): every pair, two at a time
        pm         = 0u;
        uint32_t k = 0;
        for (; k + 2 <= np; k += 2)
        {
            const PairScaled p0 = pair_scaled<ORG>(r, bvh.fan_pairs, k, org_tab), p1 = pair_scaled<ORG>(r, bvh.fan_pairs, k + 1, org_tab);
            float            m01, m23;
            uint32_t         i01, i23;
            pair_cand(p0, m01, i01);
            pair_cand(p1, m23, i23);
            const bool     p  = m23 < m01;
            const float    m  = p ? m23 : m01;
            const uint32_t im = p ? i23 : i01;
            const bool     better = m < best_t;
            best_t   = better ? m : best_t;
            best_gid = better ? im : best_gid;
        }
        if (k < np)
        {
            const PairScaled p0 = pair_scaled<ORG>(r, bvh.fan_pairs, k, org_tab);
            float            m;
            uint32_t         im;
            pair_cand(p0, m, im);
            const bool better = m < best_t;
            best_t   = better ? m : best_t;
            best_gid = better ? im : best_gid;
        }
    }
    while (pm != 0u)
    {
        const uint32_t k0 = (uint32_t)__builtin_ctz(pm);
        pm &= pm - 1u;
        if (pm != 0u)
        {
            const uint32_t k1 = (uint32_t)__builtin_ctz(pm);
            pm &= pm - 1u;
            const PairScaled p0 = pair_scaled<ORG>(r, bvh.fan_pairs, k0, org_tab), p1 = pair_scaled<ORG>(r, bvh.fan_pairs, k1, org_tab);
            float            m01, m23;
            uint32_t         i01, i23;
            pair_cand(p0, m01, i01);
            pair_cand(p1, m23, i23);
            const bool     p  = m23 < m01;
            const float    m  = p ? m23 : m01;
            const uint32_t im = p ? i23 : i01;
            const bool     better = m < best_t;
            best_t   = better ? m : best_t;
            best_gid = better ? im : best_gid;
        }
        else
        {
            const PairScaled p0 = pair_scaled<ORG>(r, bvh.fan_pairs, k0, org_tab);
            float            m;
            uint32_t         im;
            pair_cand(p0, m, im);
            const bool better = m < best_t;
            best_t   = better ? m : best_t;
            best_gid = better ? im : best_gid;
        }
    }
    // ---- unpaired triangles ----
    const uint32_t ns = bvh.fan_single_count;
    if (ns)
    {
        float    st = r.tmax;
        uint32_t si = kInvalidId;
#pragma unroll 2
        for (uint32_t j = 0; j < ns; ++j)
        {
            float4 t0, t1, t2, t3;
            load_const_tri(bvh.fan_singles, j, t0, t1, t2, t3);
            const float c      = cand(tri_scaled(r, t0, t1, t2));
            const bool  better = c < st;
            st = better ? c : st;
            si = better ? f2u(t3.x) : si;
        }
        const bool better = (st < best_t) | ((st == best_t) & (si < best_gid));
        best_t   = better ? st : best_t;
        best_gid = better ? si : best_gid;
    }
    if (best_gid != kInvalidId)
    {
        const float4*   rec = rec_tab + 4 * (size_t)best_gid;
        const TriScaled s   = tri_scaled(r, rec[0], rec[1], rec[2]);
        const float     inv = rcp_c(s.det);
        best_u = s.U * inv, best_v = s.V * inv;
    }
}

// Everything of the occlusion test that depends on the ray's DIRECTION and the triangle only: sign mask of d.n, |d.n| and the two
// interval bounds tmin * |d.n|, tmax * |d.n|.  The reference model's shadow rays of one frame share the direction (the frame's
// light) and tmin / tmax are constants, so k_trace_any computes these once per (frame slot, fan pair) and workgroup -- the same
// operations on the same operands as pair_scaled() + the occlusion test, hence the same bits -- and the pair loop reads them from
// LDS: ten vector instructions fewer per pair.
struct PairPre
{
    float4 a, b;  // per triangle: (asfloat(sign mask), det, tmin * det, tmax * det)
};
__device__ __forceinline__ float4 tri_pre(const v3 d, const v3 n, float tmin, float tmax)
{
    const float    ddn = dot3(d, n);
    const uint32_t s   = f2u(ddn) & 0x80000000u;
    const float    det = fabsf(ddn);
    return make_float4(u2f(s), det, tmin * det, tmax * det);
}
// pair k of the record list against a ray whose direction-dependent part comes from the table
__device__ __forceinline__ bool pair_occludes_pre(const Ray& r, const float4* base, uint32_t k, const float4 pa, const float4 pb)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const RawPair ConstPair;
    const RawPair p = ((const ConstPair*)base)[k];
#else
    RawPair p;
    for (int i = 0; i < 20; ++i) p.f[i] = reinterpret_cast<const float*>(base)[20 * k + i];
#endif
    const v3 v0 = mk3(p.f[0], p.f[1], p.f[2]), e1 = mk3(p.f[3], p.f[4], p.f[5]), e2 = mk3(p.f[6], p.f[7], p.f[8]),
             e3 = mk3(p.f[9], p.f[10], p.f[11]), na = mk3(p.f[12], p.f[13], p.f[14]), nb = mk3(p.f[15], p.f[16], p.f[17]);
    const v3    tvec = r.o - v0;
    const v3    q    = cross3(tvec, r.d);
    const float e2q  = dot3(e2, q);
    bool        hit;
    {
        const uint32_t s = f2u(pa.x);
        const float    U = u2f(f2u(e2q) ^ (s ^ 0x80000000u)), V = u2f(f2u(dot3(e1, q)) ^ s), T = u2f(f2u(dot3(tvec, na)) ^ (s ^ 0x80000000u));
        hit = (U >= 0.0f) & (V >= 0.0f) & (U + V <= pa.y) & (T > pa.z) & (T < pa.w);
    }
    {
        const uint32_t s = f2u(pb.x);
        const float    U = u2f(f2u(dot3(e3, q)) ^ (s ^ 0x80000000u)), V = u2f(f2u(e2q) ^ s), T = u2f(f2u(dot3(tvec, nb)) ^ (s ^ 0x80000000u));
        hit |= (U >= 0.0f) & (V >= 0.0f) & (U + V <= pb.y) & (T > pb.z) & (T < pb.w);
    }
    return hit;
}

// det == 0 needs no test: then U = V = 0 is the only way past the first three conditions and 0 < T < 0 rejects.
// PRE: pre_row is this lane's row of the (frame slot, pair) table (see PairPre)
// NEE: the EXT model's next-event rays walk the list whose tail holds the pairs that cannot occlude them (BvhDev::fan_pairs_nee)
template <bool PRE = false, bool NEE = false>
__device__ __forceinline__ bool exhaustive_any(const BvhDev& bvh, const Ray& r, const float4* pre_row = nullptr)
{
    auto occl = [&](const TriScaled& s) {
        return (s.U >= 0.0f) & (s.V >= 0.0f) & (s.U + s.V <= s.det) & (s.T > r.tmin * s.det) & (s.T < r.tmax * s.det);
    };
    bool                hit  = false;
    const uint32_t      np   = NEE ? bvh.fan_pair_nee_count : bvh.fan_pair_count;
    const float4* const list = NEE ? bvh.fan_pairs_nee : bvh.fan_pairs;
#pragma unroll 2
    for (uint32_t k = 0; k < np; ++k)
    {
        if (PRE)
            hit |= pair_occludes_pre(r, list, k, pre_row[2 * k], pre_row[2 * k + 1]);
        else
        {
            const PairScaled p = pair_scaled(r, list, k);
            hit |= occl(p.a) | occl(p.b);
        }
    }
    const uint32_t ns = bvh.fan_single_count;
#pragma unroll 2
    for (uint32_t j = 0; j < ns; ++j)
    {
        float4 t0, t1, t2, t3;
        load_const_tri(bvh.fan_singles, j, t0, t1, t2, t3);
        hit |= occl(tri_scaled(r, t0, t1, t2));
    }
    return hit;
}

// STACK == 0 selects the exhaustive small-scene path.
template <int STACK>
__device__ __forceinline__ void trace_closest_any_size(const BvhDev& bvh, const Ray& r, uint32_t* stack, float& t, float& u, float& v,
                                                       uint32_t& gid)
{
    if constexpr (STACK == 0)
        exhaustive_closest(bvh, bvh.tris_by_id, r, t, u, v, gid);
    else
        traverse_closest<STACK>(bvh, r, stack, t, u, v, gid);
}
template <int STACK>
__device__ __forceinline__ bool trace_any_any_size(const BvhDev& bvh, const Ray& r, uint32_t* stack, const float4* pre_row = nullptr)
{
    if constexpr (STACK == 0)
        return pre_row ? exhaustive_any<true>(bvh, r, pre_row) : exhaustive_any<false>(bvh, r);  // wave-uniform choice
    else
        return traverse_any<STACK>(bvh, r, stack);
}

// Workgroups per CU the stack kernels are register-allocated for: what their LDS stacks allow.  A workgroup's stack is
// entries x 256 lanes x 4 B; of 32-KB stacks four fit into the CU's 160 KB beside the runtime's own share, of 24-KB ones six.
// (0 = no hint: the small-scene kernels keep the compiler's default allocation; a hint of 8 squeezed k_trace_any<0> from 44 to
// 35 VGPRs and cost 9 %.)
constexpr int stack_residency(int stack_entries) { return stack_entries == 0 ? 0 : (stack_entries <= 24 ? 6 : (stack_entries <= 32 ? 4 : 2)); }

// One workgroup per four 64-pixel groups of a frame slot (blockIdx.y): camera rays are coherent but their cost varies strongly
// over the image, and a persistent grid with static slots left long tails here (4.1 -> 5.4 ms on the 262 k-triangle scene).
// The grid is therefore far larger than the wide traversal's spill area: binary traversal (wide8_ok cleared by the launcher).
template <int STACK>
__global__ __launch_bounds__(kBlock, stack_residency(STACK)) void k_trace_primary(BvhDev bvh, CameraDev cam, ScreenDev screen, const FrameConst* frames,
                                                          float4* hits)
{
    __shared__ uint32_t lds_stack[(STACK ? STACK : 1) * (STACK ? kBlock : 1)];
    uint32_t*           stack  = lds_stack + threadIdx.x;
    const uint32_t      slot   = blockIdx.y;
    const FrameConst    fc     = frames[slot];
    const uint32_t      chunks = screen.pixels_padded >> 6;
    for (uint32_t chunk = wave_global_id(); chunk < chunks; chunk += wave_total())
    {
        const uint32_t pl = chunk * 64 + (threadIdx.x & 63u);
        uint32_t       x, y;
        float          t = kPrimaryFar, u = 0.0f, v = 0.0f;
        uint32_t       gid = kInvalidId;
        if (local_pixel_to_xy(screen, pl, x, y))
        {
            const Ray r = make_ray(mk3(cam.position[0], cam.position[1], cam.position[2]), primary_dir(cam, screen, fc, x, y), 0.0f,
                                   kPrimaryFar);
            trace_closest_any_size<STACK>(bvh, r, stack, t, u, v, gid);
        }
        hits[(size_t)slot * screen.pixels_padded + pl] = make_float4(u, v, u2f(gid), t);
    }
}

// Packet traversal for camera rays.  The 64 rays of a chunk are the 8x8 pixels of one screen tile: they share an origin and
// nearly a direction, so the wave walks ONE node sequence with ONE stack (wave-uniform, in LDS) and fetches nodes and triangle
// records through the scalar cache; a child is entered when any lane's ray hits its box (each lane prunes with its own
// closest hit so far), nearer child first by majority.  Per-lane traversal is bound by the texture-address unit here -- seven
// 16-B loads per lane and step, 64 lanes, all to the same address: ~112 TA cycles per step and wave -- while this form issues no
// vector loads at all (262 k-triangle scene: 3.7 -> 2.4 ms; what then bounds it is instruction issue -- ~3 900 vector and ~1 800
// scalar instructions per packet, docs/experiments.md (85) -- not, as assumed until round 6, the chain of scalar fetches).  The same walk on the wide view of the tree, four boxes per step ordered by the packet's
// first live lane, was slower: 3.2 ms, the scalar sorting costs more than the halved fetch chain saves).
// The hit rule is visit-order independent (minimum t, ties to the lower id), so the result is bit-identical.
// wstack: this wave's kPacketStack entries (the tree depth is checked on the host against the same 64).
constexpr uint32_t kPacketStack = 64;
// One box of a packet step, written for what the step is bound by -- vector ISSUE (round 6: 3 880 vector instructions per packet at
// ~3 cycles each fill the SIMD's time; docs/experiments.md (85)).  Near and far plane as slab() picks them: the plane distance is
// monotonic in the plane (rounding is monotonic), increasing for inv > 0 and decreasing for inv < 0, so min / max of a slab's two
// distances IS the distance picked by the sign of inv -- a full-rate v_bitop3 select with the ray's sign word instead of a half-rate
// min / max.  Where slab()'s min / max drop a NaN (0 * inf) the select keeps it, and the max / min behind it drop it: only wider.
__device__ __forceinline__ float packet_sel(uint32_t m, float a, float b) { return u2f(__builtin_amdgcn_bitop3_b32(m, f2u(a), f2u(b), 0xca)); }
// v_max3 / v_min3 / v_max / v_min as the hardware has them.  Through fmaxf / fminf the compiler first canonicalises every operand it
// cannot prove quiet (four of the selected words per box, 4 cycles each): a signalling NaN would pass through v_max instead of being
// dropped.  None can occur here -- every operand is the result of an arithmetic instruction (NaNs from 0 * inf are quiet), a select
// between two such results, or a constant -- so the hardware forms drop NaNs exactly as fmaxf / fminf do.
__device__ __forceinline__ float packet_max3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float packet_min3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float packet_max(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float packet_min(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// Entry distance and (widened) exit distance of one box: the box is hit when tn <= tf.
// Plane distances as ONE fused operation each, t = fma(plane, inv, -(o * inv)), instead of slab()'s subtraction and product (12 of a
// step's vector instructions).  In space the crossing moves by at most u |o| (the rounding of o * inv) + u |plane - o| (the fma's),
// u = 2^-24, against slab()'s 2 u |plane - o|: both sit two orders of magnitude inside the boxes' padding of 1e-5 max(1, |coordinate|)
// for a camera within ~80 scene sizes of the scene (intersection contract), and the hit rule never looks at boxes.
// With inv = +-inf (a zero direction component) the fma can give inf - inf = NaN where slab() gave +-inf; the max / min drop it:
// the axis is ignored, which is wider.
// OCT 0 .. 7: the packet's 64 rays agree in the signs of their direction (all but the tiles a zero of a component runs through) and
// OCT holds them (bit 0: 1 / d.x negative, bit 1: y, bit 2: z): which plane is the near one is then known when the code is compiled,
// and the walk exists once per octant -- no select at all, vector or scalar (a scalar select per plane, 12 more scalar instructions
// per step, would only move the work to the unit that is second-busiest; (85)).  OCT 8: mixed signs, per-lane selects.
struct PacketSigns
{
    uint32_t mx, my, mz;  // per lane: all ones where 1 / d is negative
};
template <int OCT>
__device__ __forceinline__ void packet_slab(const Ray& r, v3 noi, const PacketSigns& g, float lox, float loy, float loz, float hix, float hiy,
                                            float hiz, float tfar, float& tn, float& tf)
{
    if constexpr (OCT < 8)
    {
        const float nx = (OCT & 1) ? hix : lox, ny = (OCT & 2) ? hiy : loy, nz = (OCT & 4) ? hiz : loz;
        const float fx = (OCT & 1) ? lox : hix, fy = (OCT & 2) ? loy : hiy, fz = (OCT & 4) ? loz : hiz;
        tn = packet_max(packet_max3(fmaf(nx, r.inv.x, noi.x), fmaf(ny, r.inv.y, noi.y), fmaf(nz, r.inv.z, noi.z)), r.tmin);
        tf = packet_min(packet_min3(fmaf(fx, r.inv.x, noi.x), fmaf(fy, r.inv.y, noi.y), fmaf(fz, r.inv.z, noi.z)), tfar) * 1.0000004f;
    }
    else
    {
        const float ax = fmaf(lox, r.inv.x, noi.x), bx = fmaf(hix, r.inv.x, noi.x);
        const float ay = fmaf(loy, r.inv.y, noi.y), by = fmaf(hiy, r.inv.y, noi.y);
        const float az = fmaf(loz, r.inv.z, noi.z), bz = fmaf(hiz, r.inv.z, noi.z);
        tn = packet_max(packet_max3(packet_sel(g.mx, bx, ax), packet_sel(g.my, by, ay), packet_sel(g.mz, bz, az)), r.tmin);
        tf = packet_min(packet_min3(packet_sel(g.mx, ax, bx), packet_sel(g.my, ay, by), packet_sel(g.mz, az, bz)), tfar) * 1.0000004f;
    }
}

#if defined(CAP_PACKET_V1)
// the round-2 form of the walk, kept for A/B runs (tools/build_variant.sh packetv1 -DCAP_PACKET_V1)
__device__ __forceinline__ void traverse_closest_packet(const BvhDev& bvh, const Ray& r, bool alive, uint32_t* wstack, float& best_t,
                                                        float& best_u, float& best_v, uint32_t& best_gid)
{
    best_t = r.tmax, best_u = 0.0f, best_v = 0.0f, best_gid = kInvalidId;
    int      node = bvh.root;
    uint32_t sp   = 0;
    while (true)
    {
        node = __builtin_amdgcn_readfirstlane(node);
        bool pop = true;
        if (node >= 0)
        {
            float4 q0, q1, q2, q3;
            load_const_tri(bvh.nodes, (uint32_t)node, q0, q1, q2, q3);
            float      tn0, tn1;
            const bool h0 = alive && slab(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, best_t, tn0);
            const bool h1 = alive && slab(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, best_t, tn1);
            const int  c0 = (int)f2u(q3.z), c1 = (int)f2u(q3.w);
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
            if (m0 != 0ull && m1 != 0ull)
            {
                // lanes that hit both vote for the nearer child, the others for the one they hit
                const unsigned long long first1 = __ballot(h1 && (!h0 || tn1 < tn0));
                const bool               swap   = 2 * __popcll(first1) > __popcll(m0 | m1);
                if (sp < kPacketStack) wstack[sp++] = (uint32_t)(swap ? c0 : c1);
                node = swap ? c1 : c0;
                pop  = false;
            }
            else if ((m0 | m1) != 0ull)
            {
                node = m0 ? c0 : c1;
                pop  = false;
            }
        }
        else
        {
            const uint32_t code = (uint32_t)~node, first = code & kLeafFirstMask, last = first + (code >> kLeafCountShift);
            for (uint32_t leaf = first; leaf <= last; ++leaf)
            {
                float4 t0, t1, t2, t3;
                load_const_tri(bvh.tris, leaf, t0, t1, t2, t3);
                float t, u, v;
                if (alive && tri_test(r, t0, t1, t2, t, u, v))
                {
                    const uint32_t gid = f2u(t3.x);
                    if (t < best_t || (t == best_t && gid < best_gid)) best_t = t, best_u = u, best_v = v, best_gid = gid;
                }
            }
        }
        if (pop)
        {
            if (sp == 0) break;
            node = (int)wstack[--sp];
        }
    }
}
#else
template <int OCT>
__device__ __forceinline__ void packet_walk(const BvhDev& bvh, const Ray& r, v3 noi, const PacketSigns& g, unsigned long long alive_mask,
                                            uint32_t* wstack, float& best_t, float& best_u, float& best_v, uint32_t& best_gid)
{
    int      node = bvh.root;
    uint32_t sp   = 0;
    while (true)
    {
        node = __builtin_amdgcn_readfirstlane(node);
        bool pop = true;
        if (node >= 0)
        {
            float4 q0, q1, q2, q3;
            load_const_tri(bvh.nodes, (uint32_t)node, q0, q1, q2, q3);
            // (no `alive &&` around the tests: a lane without a pixel computes on its dummy ray; the branches cost scalar issue slots
            // and kept the two boxes from being scheduled together)
            float tn0, tn1, tf0, tf1;
            packet_slab<OCT>(r, noi, g, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, best_t, tn0, tf0);
            packet_slab<OCT>(r, noi, g, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, best_t, tn1, tf1);
            const int c0 = (int)f2u(q3.z), c1 = (int)f2u(q3.w);
            // Lanes that hit both vote for the nearer child, the others for the one they hit; the majority's child is entered first.
            // One rule for all three cases: only child 1 hit -> every voter says 1, only child 0 -> nobody does.
            // (the compares' lane masks, AND-ed with the mask of lanes that have a pixel: the ballot of a computed bool costs two
            // half-rate vector instructions, the ballot of a compare none)
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(tn0 <= tf0) & alive_mask, m1 = __builtin_amdgcn_ballot_w64(tn1 <= tf1) & alive_mask;
            const unsigned long long first1 = m1 & (~m0 | __builtin_amdgcn_ballot_w64(tn1 < tn0));
            const bool               swap   = 2 * __popcll(first1) > __popcll(m0 | m1);
            if ((m0 | m1) != 0ull)
            {
                // (no depth guard: the host refuses trees deeper than kPacketStack for this walk; the index is masked for safety)
                if (m0 != 0ull && m1 != 0ull) wstack[(sp++) & (kPacketStack - 1u)] = (uint32_t)(swap ? c0 : c1);
                node = swap ? c1 : c0;
                pop  = false;
            }
        }
        else
        {
            const uint32_t code = (uint32_t)~node, first = code & kLeafFirstMask, last = first + (code >> kLeafCountShift);
            for (uint32_t leaf = first; leaf <= last; ++leaf)
            {
                float4 t0, t1, t2, t3;
                load_const_tri(bvh.tris, leaf, t0, t1, t2, t3);
                float t, u, v;
                // (a lane without a pixel walks its dummy ray: its result is dropped below.  Skipping the reciprocal and the interval test
                // when no lane of the wave is inside the triangle measured nothing: (85))
                if (tri_test(r, t0, t1, t2, t, u, v))
                {
                    const uint32_t gid = f2u(t3.x);
                    if (t < best_t || (t == best_t && gid < best_gid)) best_t = t, best_u = u, best_v = v, best_gid = gid;
                }
            }
        }
        if (pop)
        {
            if (sp == 0) break;
            node = (int)wstack[--sp];
        }
    }
}

__device__ __forceinline__ void traverse_closest_packet(const BvhDev& bvh, const Ray& r_in, bool alive, uint32_t* wstack, float& best_t,
                                                        float& best_u, float& best_v, uint32_t& best_gid)
{
    // 1 / d for the BOX tests from v_rcp_f32 (1 ulp; d = +-0 gives +-inf like the division): three IEEE divisions per ray less.  Like
    // the wide view's make_wide_ray: the boxes' padding covers it and the hit rule never looks at boxes (tri_test does not use inv).
    Ray r = r_in;
    r.inv = mk3(__builtin_amdgcn_rcpf(r.d.x), __builtin_amdgcn_rcpf(r.d.y), __builtin_amdgcn_rcpf(r.d.z));
    // (The camera position is wave-uniform and lives in scalar registers, like the triangle records: three moves per triangle test.
    // Pinning it into vector registers saves them and costs more in spills at this kernel's 64 registers: + 1 %.)
    const v3 noi = mk3(-(r.o.x * r.inv.x), -(r.o.y * r.inv.y), -(r.o.z * r.inv.z));
    PacketSigns g;
    g.mx = (uint32_t)((int)f2u(r.inv.x) >> 31), g.my = (uint32_t)((int)f2u(r.inv.y) >> 31), g.mz = (uint32_t)((int)f2u(r.inv.z) >> 31);
    const unsigned long long alive_mask = __builtin_amdgcn_ballot_w64(alive);
    const unsigned long long bx = __builtin_amdgcn_ballot_w64(g.mx != 0u), by = __builtin_amdgcn_ballot_w64(g.my != 0u),
                             bz = __builtin_amdgcn_ballot_w64(g.mz != 0u);
    // every lane of the wave is active here (the chunk loop is wave-uniform), so "all 64 agree" is a ballot of 0 or of all ones
    const bool     uniform = alive_mask == ~0ull && (bx == 0ull || bx == ~0ull) && (by == 0ull || by == ~0ull) && (bz == 0ull || bz == ~0ull);
    const uint32_t oct     = uniform ? ((bx != 0ull ? 1u : 0u) | (by != 0ull ? 2u : 0u) | (bz != 0ull ? 4u : 0u)) : 8u;
    best_t = r.tmax, best_u = 0.0f, best_v = 0.0f, best_gid = kInvalidId;
    switch (oct)
    {
#define CAP_PACKET_CASE(K) case K: packet_walk<K>(bvh, r, noi, g, alive_mask, wstack, best_t, best_u, best_v, best_gid); break;
        CAP_PACKET_CASE(0) CAP_PACKET_CASE(1) CAP_PACKET_CASE(2) CAP_PACKET_CASE(3) CAP_PACKET_CASE(4) CAP_PACKET_CASE(5) CAP_PACKET_CASE(6) CAP_PACKET_CASE(7)
#undef CAP_PACKET_CASE
        default: packet_walk<8>(bvh, r, noi, g, alive_mask, wstack, best_t, best_u, best_v, best_gid); break;
    }
    if (!alive) best_t = r.tmax, best_u = 0.0f, best_v = 0.0f, best_gid = kInvalidId;
}
#endif

template <int DUMMY>
__global__ __launch_bounds__(kBlock, 8) void k_trace_primary_packet(BvhDev bvh, CameraDev cam, ScreenDev screen, const FrameConst* frames,
                                                                    uint32_t n_slots, float4* hits, uint32_t* work)
{
    __shared__ uint32_t lds_wstack[(kBlock / 64) * kPacketStack];
    uint32_t*           wstack   = lds_wstack + (threadIdx.x >> 6) * kPacketStack;
    const uint32_t      cps      = screen.pixels_padded >> 6;
    const uint32_t      chunks   = cps * n_slots;
    const uint32_t      my_class = wave_global_id() % kQueueClasses;
    uint32_t            grab     = grab_issue(work, my_class);
    while (true)
    {
        const uint32_t chunk = grab_value(grab) * kQueueClasses + my_class;
        if (chunk >= chunks) break;
        grab = grab_issue(work, my_class);
        const uint32_t slot = chunk / cps;  // wave-uniform
        const uint32_t pl   = (chunk - slot * cps) * 64 + (threadIdx.x & 63u);
        uint32_t       x = 0, y = 0;
        const bool     alive = local_pixel_to_xy(screen, pl, x, y);
        const Ray      r     = make_ray(mk3(cam.position[0], cam.position[1], cam.position[2]),
                                        alive ? primary_dir(cam, screen, frames[slot], x, y) : mk3(0.f, 0.f, 1.f), 0.0f, kPrimaryFar);
        float          t, u, v;
        uint32_t       gid;
        traverse_closest_packet(bvh, r, alive, wstack, t, u, v, gid);
        hits[(size_t)slot * screen.pixels_padded + pl] = make_float4(alive ? u : 0.0f, alive ? v : 0.0f, u2f(gid), alive ? t : kPrimaryFar);
    }
}

// Camera rays as an "identity queue" for the wide closest-hit kernel (dense scenes, see launch_raygen_identity): entry i is the ray
// of (frame slot, local pixel) = (i / Ppad, i % Ppad), so the hit record lands where the bounce-0 shade stage looks for it.
// Padding lanes of partial tiles get an empty interval.  The 64 sub-queue counters are written here too: class k owns entries
// [k * capacity, min((k + 1) * capacity, total)).
__global__ __launch_bounds__(kBlock) void k_raygen_identity(CameraDev cam, ScreenDev screen, const FrameConst* frames, uint32_t n_slots, float4* org,
                                                            float4* dir, uint32_t* count, uint32_t capacity)
{
    const uint32_t Ppad = screen.pixels_padded, total = n_slots * Ppad;
    if (blockIdx.x == 0 && threadIdx.x < kQueueClasses)
    {
        const uint64_t begin = (uint64_t)threadIdx.x * capacity;
        count[threadIdx.x * kCounterStride] = begin < total ? (uint32_t)(total - begin < capacity ? total - begin : capacity) : 0u;
    }
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock)
    {
        const uint32_t slot = i / Ppad, pl = i - slot * Ppad;
        uint32_t       x = 0, y = 0;
        const bool     alive = local_pixel_to_xy(screen, pl, x, y);
        const v3       d     = alive ? primary_dir(cam, screen, frames[slot], x, y) : mk3(0.f, 0.f, 1.f);
        org[i] = make_float4(cam.position[0], cam.position[1], cam.position[2], 0.0f);
        dir[i] = make_float4(d.x, d.y, d.z, alive ? kPrimaryFar : 0.0f);
    }
}

void launch_raygen_identity(const LaunchCfg& cfg, const CameraDev& cam, const ScreenDev& screen, const FrameConst* frames, uint32_t n_slots,
                            const RayQueue& q)
{
    const uint32_t total = n_slots * screen.pixels_padded;
    uint32_t       g     = (total + kBlock - 1) / kBlock;
    const uint32_t cap   = (cfg.cu_count ? cfg.cu_count : 256u) * 8u;
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    hipLaunchKernelGGL(k_raygen_identity, dim3(g), dim3(kBlock), 0, cfg.stream, cam, screen, frames, n_slots, q.org_tmin, q.dir_tmax, q.count,
                       q.class_capacity);
}

template <int STACK>
__global__ __launch_bounds__(kBlock, stack_residency(STACK)) void k_trace_closest(BvhDev bvh, RayQueue q, float4* hits)
{
    __shared__ uint32_t lds_stack[(STACK ? STACK : 1) * (STACK ? kBlock : 1)];
    uint32_t*           stack  = lds_stack + threadIdx.x;
    const uint32_t      slots  = (q.class_capacity >> 6) * kQueueClasses;
    for (uint32_t cs = wave_global_id(); cs < slots; cs += wave_total())
    {
        uint32_t i, klass;
        if (queue_chunk(q.count, q.class_capacity, cs, threadIdx.x & 63u, i, klass))
        {
            const float4 a = q.org_tmin[i], b = q.dir_tmax[i];
            const Ray    r = make_ray(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w, b.w);
            float        t, u, v;
            uint32_t     gid;
            trace_closest_any_size<STACK>(bvh, r, stack, t, u, v, gid);
            hits[i] = make_float4(u, v, u2f(gid), t);
        }
    }
}

// An unoccluded ray adds its contribution to its path's plane entry by load-add-store: the path is the entry's only writer
// within a launch and launches are ordered on the stream, so the sums are the same IEEE additions in the same order whichever
// way they are carried out.  RMW says WHEN the entry is loaded.  true: up front with the ray (next-event rays of the EXT model,
// ~90 % unoccluded: the load's latency hides behind the test; three float atomics per ray instead made the L2 atomic rate the
// kernel's bound, 31 -> 10.8 ms per step).  false: after the test, by the few lanes that need it (the reference's directional
// light inside the box: ~5 % unoccluded; an up-front load costs 7.05 -> 7.98 ms, float atomics 6.6 ms, this 5.8 ms).
// Entry formats.  RMW (EXT model, next-event rays): (origin, tmin) (direction, tmax) (contribution, path id), 48 B.
// !RMW (reference model): (origin, path id) (contribution, -), 32 B; direction = the light of the path's frame (LDS copy of the
// batch's frame constants), tmin / tmax = kRayEps / kRayFar (lighting.h:39-47).
// k_trace_any's table of PairPre entries lives in dynamic LDS (pre_table_bytes(), passed at launch; 0 = no table): rows of
// 2 * pairs + 1 float4 per frame slot -- the extra one shifts the banks, so that lanes reading the same pair of different frame
// slots do not collide.  8 KB for the headline's 16-frame batches, 34 KB for a 64-frame batch of a sharded run.
constexpr uint32_t kPreTableMaxBytes = 40u << 10;
static inline uint32_t pre_table_bytes(uint32_t n_slots, uint32_t pairs)
{
    const uint64_t b = (uint64_t)n_slots * (2u * pairs + 1u) * sizeof(float4);
    return (pairs && n_slots <= kMaxFrameSlots && b <= kPreTableMaxBytes) ? (uint32_t)b : 0u;
}
template <int STACK, bool RMW>
__global__ __launch_bounds__(kBlock, stack_residency(STACK)) void k_trace_any(BvhDev bvh, ShadowQueue q, float4* target, uint32_t pixels_padded, uint32_t n_slots,
                                                      uint64_t* guard, uint32_t* work, const FrameConst* frames, uint32_t pre_bytes)
{
    extern __shared__ float4 lds_pre[];
    __shared__ uint32_t lds_stack[(STACK ? STACK : 1) * (STACK ? kBlock : 1)];
    __shared__ float4   lds_light[RMW ? 1 : kMaxFrameSlots];
    // reference model on the small-scene path: the direction-dependent part of the pair tests per (frame slot, pair), see PairPre
    constexpr bool      DDN = STACK == 0 && !RMW;
    uint32_t*           stack    = lds_stack + threadIdx.x;
    const uint32_t      pre_row  = 2u * bvh.fan_pair_count + 1u;  // float4 per frame slot
    const bool          use_ddn  = DDN && pre_bytes != 0u;
    if (!RMW)
    {
        for (uint32_t k = threadIdx.x; k < n_slots && k < kMaxFrameSlots; k += kBlock)
            lds_light[k] = make_float4(frames[k].light_dir[0], frames[k].light_dir[1], frames[k].light_dir[2], 0.f);
        if (use_ddn)
        {
            const uint32_t np = bvh.fan_pair_count;
            const float*   fp = reinterpret_cast<const float*>(bvh.fan_pairs);
            for (uint32_t e = threadIdx.x; e < n_slots * np; e += kBlock)
            {
                const uint32_t slot = e / np, k = e - slot * np;
                const v3       d    = mk3(frames[slot].light_dir[0], frames[slot].light_dir[1], frames[slot].light_dir[2]);
                const float*   rec  = fp + 20 * (size_t)k;  // (v0, e1, e2, e3, nA, nB, id, 0)
                lds_pre[slot * pre_row + 2 * k]     = tri_pre(d, mk3(rec[12], rec[13], rec[14]), kRayEps, kRayFar);
                lds_pre[slot * pre_row + 2 * k + 1] = tri_pre(d, mk3(rec[15], rec[16], rec[17]), kRayEps, kRayFar);
            }
        }
        __syncthreads();
    }
    // Reference model on the wide tree: every shadow ray of the launch points at its frame's light, and the frames of a batch differ by
    // a jitter -- one direction octant for the whole launch unless a component of the light direction is (almost) zero.  Told here from
    // the LDS copy of the lights (sign bits, as make_wide_ray reads them off 1 / d); the traversal then runs in the copy compiled for
    // that octant (traverse_any8<.., OCT>: near / far planes and the visiting permutation are constants; docs/experiments.md (86)).
    uint32_t oct = 8u;  // mixed, or not this instantiation: per-lane signs
    if constexpr (!RMW && STACK == (int)kWideLdsEntries)
    {
        const float4   L  = lds_light[(threadIdx.x & 63u) < n_slots ? (threadIdx.x & 63u) : 0u];
        const uint32_t o  = (f2u(L.x) >> 31) | ((f2u(L.y) >> 31) << 1) | ((f2u(L.z) >> 31) << 2);
        const uint32_t o0 = __builtin_amdgcn_readfirstlane(o);
        if (__builtin_amdgcn_ballot_w64(o != o0) == 0ull && bvh.wide8_ok && bvh.tri_count != 0u) oct = o0;  // (all 64 lanes are active here)
    }
    // Memory round trips a chunk starts with, in order: (1) the grab issued one chunk ago (its wait also covers the previous
    // chunk's plane updates: vmcnt retires in order), (2) the queue entry.  The class's length is read once per wave (constant
    // during the launch), and the next grab is issued AFTER the entry loads, so that the wait for the entry is a counted
    // vmcnt(1) that leaves the grab in flight instead of a third round trip.
    const uint32_t my_class = wave_global_id() % kQueueClasses;
    const uint32_t lane     = threadIdx.x & 63u;
    uint32_t       n_class  = q.count[my_class * kCounterStride];
    n_class                 = n_class < q.class_capacity ? n_class : q.class_capacity;
    uint32_t       grab     = grab_issue(work, my_class);
    while (true)
    {
        const uint32_t local0 = grab_value(grab) * 64u;
        if (local0 >= n_class) break;  // past the end of this class's sub-queue
        const bool     active = local0 + lane < n_class;
        const uint32_t i      = my_class * q.class_capacity + local0 + lane;
        float4         a      = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) a = q.org_tmin[i];
        grab = grab_issue(work, my_class);
        if (active)
        {
            float4       c = make_float4(0.f, 0.f, 0.f, 0.f), cur = c;
            uint32_t     pid = 0;
            size_t       idx = 0;
            bool         good = true;
            auto         locate = [&]() {
                good = (pid >> kPidShift) < n_slots && (pid & kPidMask) < pixels_padded;
                idx  = (size_t)(pid >> kPidShift) * pixels_padded + (pid & kPidMask);
                if (!good)
                {
                    // never true for a well-formed queue; reported through CapStats::guard_* instead of faulting
                    atomicAdd((unsigned long long*)guard + 2, 1ull);
                    guard[3] = ((uint64_t)i << 32) | pid;
                }
            };
            Ray r;
            if (RMW)
            {
                const float4 b = q.dir_tmax[i];
                r   = make_ray(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w, b.w);
                c   = q.contrib_pid[i];
                pid = f2u(c.w);
                locate();
                if (good) cur = target[idx];
            }
            else
            {
                pid = f2u(a.w);
                locate();
                const float4 L = lds_light[good ? (pid >> kPidShift) : 0u];
                r = make_ray(mk3(a.x, a.y, a.z), mk3(L.x, L.y, L.z), kRayEps, good ? kRayFar : 0.0f);  // malformed entry: empty interval
            }
            const float4* row = use_ddn ? lds_pre + (good ? (pid >> kPidShift) : 0u) * pre_row : nullptr;
            if (STACK == 0) __builtin_amdgcn_s_setprio(0);
            bool occluded;
            if constexpr (!RMW && STACK == (int)kWideLdsEntries)
            {
                switch (oct)  // wave-uniform
                {
#define CAP_ANY_OCT(K) case K: occluded = traverse_any8<STACK, K>(bvh, r, stack); break;
                    // (the four octants the reference's light visits: it turns about the vertical axis, y stays positive -- lighting.h:20-33;
                    // tests/test_sponza_class_gpu.py test_light_octants_parity renders a frame of each)
                    CAP_ANY_OCT(0) CAP_ANY_OCT(1) CAP_ANY_OCT(4) CAP_ANY_OCT(5)
#undef CAP_ANY_OCT
                    default: occluded = trace_any_any_size<STACK>(bvh, r, stack, row); break;
                }
            }
            else
                occluded = trace_any_any_size<STACK>(bvh, r, stack, row);
            if (STACK == 0) __builtin_amdgcn_s_setprio(3);
            if (!occluded)
            {
                // lighting.h:57-60: unoccluded -> the contribution evaluated at shading time is added
                if (good)
                {
                    if (RMW)
                        target[idx] = make_float4(cur.x + c.x, cur.y + c.y, cur.z + c.z, cur.w);
                    else
                    {
                        c           = q.contrib_pid[i];
                        cur         = target[idx];
                        target[idx] = make_float4(cur.x + c.x, cur.y + c.y, cur.z + c.z, cur.w);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Shadow rays of the reference model on small scenes, with occluder-first ordering and wave-level compaction.
//
// An any-hit query is over at the first occluder, but the exhaustive loop is wave-uniform: it could only stop early when all 64
// rays of the wave are occluded.  Inside a box most shadow rays ARE occluded, and by very few of the triangles -- whatever lies
// farthest along the light direction (the ceiling of the Cornell box for the reference's light from above: 84 % of the shadow
// rays of the headline workload; the other 16 % leave through the open front).  So the loop is split:
//   probe   every ray is tested against the first `probe` fan pairs of an order sorted by how far along the light direction a
//           pair's centre lies (per launch, for the light of the batch's first frame);
//   rest    the rays that survived are parked in a per-wave LDS buffer; whenever 64 of them have gathered, the wave tests those 64
//           against the remaining pairs and the unpaired triangles, and adds the contributions of the unoccluded ones.
// An occlusion query's answer does not depend on the order of the tests, and each path has at most one shadow entry per launch,
// so the planes receive the same additions as in k_trace_any: bit-identical images and counters.
// ------------------------------------------------------------------------------------------------
#ifndef CAP_ANY_GRAB
#define CAP_ANY_GRAB 4  // chunk slots per grab of the probe kernel
#endif
#ifndef CAP_ANY_PROBE
#define CAP_ANY_PROBE 1
#endif
constexpr uint32_t kSurvivorCap = 128;  // per wave: <= 63 parked + <= 64 new
__global__ __launch_bounds__(kBlock) void k_trace_any_small(BvhDev bvh, ShadowQueue q, float4* target, uint32_t pixels_padded, uint32_t n_slots,
                                                           uint64_t* guard, uint32_t* work, const FrameConst* frames, uint32_t probe)
{
    extern __shared__ float4 lds_pre[];  // PairPre rows per frame slot, see k_trace_any
    __shared__ float4   lds_light[kMaxFrameSlots];
    __shared__ float    lds_score[kExhaustiveMax / 2];
    __shared__ uint32_t lds_order[kExhaustiveMax / 2];
    __shared__ float4   lds_surv[(kBlock / 64) * kSurvivorCap];    // (origin, path id) of parked rays
    __shared__ uint32_t lds_surv_i[(kBlock / 64) * kSurvivorCap];  // their queue entries
    const uint32_t np      = bvh.fan_pair_count;  // <= kExhaustiveMax / 2
    const uint32_t pre_row = 2u * np + 1u;
    const float*   fp      = reinterpret_cast<const float*>(bvh.fan_pairs);
    for (uint32_t k = threadIdx.x; k < n_slots && k < kMaxFrameSlots; k += kBlock)
        lds_light[k] = make_float4(frames[k].light_dir[0], frames[k].light_dir[1], frames[k].light_dir[2], 0.f);
    for (uint32_t e = threadIdx.x; e < n_slots * np; e += kBlock)
    {
        const uint32_t slot = e / np, k = e - slot * np;
        const v3       d    = mk3(frames[slot].light_dir[0], frames[slot].light_dir[1], frames[slot].light_dir[2]);
        const float*   rec  = fp + 20 * (size_t)k;  // (v0, e1, e2, e3, nA, nB, id, 0)
        lds_pre[slot * pre_row + 2 * k]     = tri_pre(d, mk3(rec[12], rec[13], rec[14]), kRayEps, kRayFar);
        lds_pre[slot * pre_row + 2 * k + 1] = tri_pre(d, mk3(rec[15], rec[16], rec[17]), kRayEps, kRayFar);
    }
    if (threadIdx.x < np)
    {
        const float* rec = fp + 20 * (size_t)threadIdx.x;
        const v3     L   = mk3(frames[0].light_dir[0], frames[0].light_dir[1], frames[0].light_dir[2]);
        const v3     v0  = mk3(rec[0], rec[1], rec[2]);
        float        sc  = dot3(v0, L);  // sum over the quad's four vertices = 4 x its centre's reach
        for (int e = 0; e < 3; ++e) sc += dot3(v0 + mk3(rec[3 + 3 * e], rec[4 + 3 * e], rec[5 + 3 * e]), L);
        lds_score[threadIdx.x] = sc;
    }
    __syncthreads();
    if (threadIdx.x < np)
    {
        // rank sort: position = pairs whose centre lies farther along the light (ties by index)
        const float sc   = lds_score[threadIdx.x];
        uint32_t    rank = 0;
        for (uint32_t j = 0; j < np; ++j)
        {
            const float o = lds_score[j];
            rank += (o > sc || (o == sc && j < threadIdx.x)) ? 1u : 0u;
        }
        lds_order[rank] = threadIdx.x;
    }
    __syncthreads();
    const uint32_t  n_probe = probe < np ? probe : np;
    const uint32_t  lane    = threadIdx.x & 63u;
    float4* const   surv    = lds_surv + (threadIdx.x >> 6) * kSurvivorCap;
    uint32_t* const surv_i  = lds_surv_i + (threadIdx.x >> 6) * kSurvivorCap;
    uint32_t        surv_n  = 0;  // wave-uniform

    // the rest of the tests for the parked rays [first, first + count), count <= 64
    auto finish = [&](uint32_t first, uint32_t count) {
        if (lane < count)
        {
            const float4   a    = surv[first + lane];
            const uint32_t i    = surv_i[first + lane];
            const uint32_t pid  = f2u(a.w);
            const bool     good = (pid >> kPidShift) < n_slots && (pid & kPidMask) < pixels_padded;  // counted once, in the probe
            const uint32_t slot = good ? (pid >> kPidShift) : 0u;
            const float4   L    = lds_light[slot];
            const Ray      r    = make_ray(mk3(a.x, a.y, a.z), mk3(L.x, L.y, L.z), kRayEps, good ? kRayFar : 0.0f);
            const float4*  row  = lds_pre + slot * pre_row;
            bool           hit  = false;
            for (uint32_t j = n_probe; j < np; ++j)
            {
                const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_order[j]);  // wave-uniform: scalar-cache record
                hit |= pair_occludes_pre(r, bvh.fan_pairs, k, row[2 * k], row[2 * k + 1]);
            }
            const uint32_t ns = bvh.fan_single_count;
            for (uint32_t j = 0; j < ns; ++j)
            {
                float4 t0, t1, t2, t3;
                load_const_tri(bvh.fan_singles, j, t0, t1, t2, t3);
                hit |= tri_occludes(r, t0, t1, t2);
            }
            if (!hit && good)
            {
                // lighting.h:57-60: unoccluded -> the contribution evaluated at shading time is added
                const size_t idx = (size_t)(pid >> kPidShift) * pixels_padded + (pid & kPidMask);
                const float4 c = q.contrib_pid[i], cur = target[idx];
                target[idx]    = make_float4(cur.x + c.x, cur.y + c.y, cur.z + c.z, cur.w);
            }
        }
    };

    const uint32_t my_class = wave_global_id() % kQueueClasses;
    uint32_t       n_class  = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.count[my_class * kCounterStride]);
    n_class                 = n_class < q.class_capacity ? n_class : q.class_capacity;
    // The probe is short (a few hundred cycles per chunk), far shorter than a returned device atomic or a queue-entry load take:
    // a grab therefore fetches kGrabChunks consecutive chunk slots of the class, the grab for the next group is issued when a
    // group is started, and within the stream of chunks the entry of the NEXT chunk is loaded before this chunk is worked on.
    constexpr uint32_t kGrabChunks = CAP_ANY_GRAB;
    auto grab_group = [&]() {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(work + my_class * kCounterStride, kGrabChunks);
        return v;
    };
    uint32_t grab     = grab_group();
    uint32_t local0   = grab_value(grab) * 64u;  // first ray of the chunk being worked on
    grab              = grab_group();
    uint32_t in_group = 0;
    float4   a_next   = make_float4(0.f, 0.f, 0.f, 0.f);
    if (local0 + lane < n_class) a_next = q.org_tmin[my_class * q.class_capacity + local0 + lane];
    while (true)
    {
        if (local0 >= n_class) break;  // past the end of this class's sub-queue (groups are handed out in order)
        const bool     active = local0 + lane < n_class;
        const uint32_t i      = my_class * q.class_capacity + local0 + lane;
        const float4   a      = a_next;
        // the next chunk: the next one of this group, or the first one of the next group
        if (++in_group == kGrabChunks)
        {
            in_group = 0;
            local0   = grab_value(grab) * 64u;
            grab     = grab_group();
        }
        else
            local0 += 64u;
        if (local0 + lane < n_class) a_next = q.org_tmin[my_class * q.class_capacity + local0 + lane];
        bool hit = false;
        if (active)
        {
            const uint32_t pid  = f2u(a.w);
            const bool     good = (pid >> kPidShift) < n_slots && (pid & kPidMask) < pixels_padded;
            if (!good)
            {
                // never true for a well-formed queue; reported through CapStats::guard_* instead of faulting
                atomicAdd((unsigned long long*)guard + 2, 1ull);
                guard[3] = ((uint64_t)i << 32) | pid;
            }
            const uint32_t slot = good ? (pid >> kPidShift) : 0u;
            const float4   L    = lds_light[slot];
            const Ray      r    = make_ray(mk3(a.x, a.y, a.z), mk3(L.x, L.y, L.z), kRayEps, good ? kRayFar : 0.0f);
            const float4*  row  = lds_pre + slot * pre_row;
            for (uint32_t j = 0; j < n_probe; ++j)
            {
                const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_order[j]);
                hit |= pair_occludes_pre(r, bvh.fan_pairs, k, row[2 * k], row[2 * k + 1]);
            }
        }
        // park the survivors
        const bool               alive = active && !hit;
        const unsigned long long m     = __ballot(alive);
        if (alive)
        {
            const uint32_t at = surv_n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            surv[at]   = a;
            surv_i[at] = i;
        }
        surv_n += (uint32_t)__popcll(m);
        wave_handoff();
        if (surv_n >= 64u)
        {
            surv_n -= 64u;
            finish(surv_n, 64u);
            wave_handoff();
        }
    }
    if (surv_n) finish(0u, surv_n);
}

// ------------------------------------------------------------------------------------------------
// LBVH traversal with lane refill.  Incoherent rays leave a wave in the stack loop for the MAXIMUM of 64 traversal lengths
// (measured on the 262 k-triangle scene: 11 of 64 lanes active on average).  Here a lane that finishes its ray takes the next
// ray of the wave's own chunk sequence, so the wave keeps its lanes busy until that sequence is exhausted.  The feed is
// wave-uniform bookkeeping (no atomics); every ray is still traced by exactly one lane and writes its own queue index.
// ------------------------------------------------------------------------------------------------
struct WaveFeed
{
    uint32_t cs, slots, stride;  // next chunk slot of this wave, slot count, slot stride (waves in the grid)
    uint32_t base, n, pos;       // current chunk: first queue index, rays in it, rays already handed out
    bool     exhausted;
};

__device__ __forceinline__ void feed_init(WaveFeed& f, uint32_t class_capacity)
{
    f.cs = wave_global_id(), f.slots = (class_capacity >> 6) * kQueueClasses, f.stride = wave_total();
    f.base = f.n = f.pos = 0;
    f.exhausted = false;
}

// Move to this wave's next non-empty chunk.  All values are wave-uniform.
__device__ __forceinline__ void feed_advance(WaveFeed& f, const uint32_t* count, uint32_t class_capacity)
{
    while (f.pos >= f.n && !f.exhausted)
    {
        if (f.cs >= f.slots)
        {
            f.exhausted = true;
            break;
        }
        const uint32_t klass = f.cs % kQueueClasses, j = f.cs / kQueueClasses;
        const uint32_t cnt   = __builtin_amdgcn_readfirstlane(count[klass * kCounterStride]);
        const uint32_t start = j * 64u;
        f.n    = cnt > start ? min(64u, cnt - start) : 0u;
        f.base = klass * class_capacity + start;
        f.pos  = 0;
        f.cs += f.stride;
    }
}

// Hands rays to idle lanes; returns this lane's new queue index or ~0u.
__device__ __forceinline__ uint32_t feed_take(WaveFeed& f, bool idle, const uint32_t* count, uint32_t class_capacity)
{
    uint32_t           mine = kInvalidId;
    unsigned long long mask = __ballot(idle);
    while (mask != 0ull)
    {
        feed_advance(f, count, class_capacity);
        if (f.exhausted) break;
        const uint32_t lane  = threadIdx.x & 63u;
        const uint32_t rank  = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        const uint32_t avail = f.n - f.pos, want = (uint32_t)__popcll(mask);
        const uint32_t take  = avail < want ? avail : want;
        if (idle && mine == kInvalidId && rank < take) mine = f.base + f.pos + rank;
        f.pos += take;
        mask = __ballot(idle && mine == kInvalidId);
    }
    return mine;
}

#ifndef CAP_REFILL_IDLE
#define CAP_REFILL_IDLE 28
#endif
#ifndef CAP_LEAF_BATCH
#define CAP_LEAF_BATCH 24
#endif
constexpr uint32_t kRefillIdle = CAP_REFILL_IDLE;  // refill once this many lanes are idle (amortises the ray-load latency over several lanes)
constexpr int      kLeafBatch  = CAP_LEAF_BATCH;  // keep running the box code while at least this many lanes are on internal nodes

template <int STACK>
__global__ __launch_bounds__(kBlock, stack_residency(STACK)) void k_trace_closest_refill(BvhDev bvh, RayQueue q, float4* hits)
{
    __shared__ uint32_t lds_stack[STACK * kBlock];
    uint32_t*           stack = lds_stack + threadIdx.x;
    WaveFeed            feed;
    feed_init(feed, q.class_capacity);
    if (bvh.tri_count == 0)
    {
        // no geometry: every queued ray misses
        for (uint32_t cs = wave_global_id(); cs < feed.slots; cs += wave_total())
        {
            uint32_t i, klass;
            if (queue_chunk(q.count, q.class_capacity, cs, threadIdx.x & 63u, i, klass)) hits[i] = make_float4(0.f, 0.f, u2f(kInvalidId), 0.f);
        }
        return;
    }
    bool     alive = false;
    Ray      r     = make_ray(mk3(0, 0, 0), mk3(0, 0, 1), 0.f, 0.f);
    float    best_t = 0.f, best_u = 0.f, best_v = 0.f;
    uint32_t best_gid = kInvalidId, out = 0;
    int      node = 0, sp = 0;
    while (true)
    {
        const uint32_t n_alive = (uint32_t)__popcll(__ballot(alive));
        if (!feed.exhausted && 64u - n_alive >= kRefillIdle)
        {
            const uint32_t i = feed_take(feed, !alive, q.count, q.class_capacity);
            if (i != kInvalidId)
            {
                const float4 a = q.org_tmin[i], b = q.dir_tmax[i];
                r      = make_ray(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w, b.w);
                best_t = r.tmax, best_u = 0.f, best_v = 0.f, best_gid = kInvalidId;
                out = i, node = bvh.root, sp = 0, alive = true;
            }
        }
        if (__ballot(alive) == 0ull) break;  // feed exhausted and every lane retired
        // while-while: as long as enough lanes sit on an internal node only the box code runs; lanes that reached a leaf wait
        // until leaves are due (few lanes left on internal nodes), then only the triangle code runs.  Every iteration pays for
        // one of the two bodies instead of both.
        const unsigned long long m_inner = __ballot(alive && node >= 0);
        const unsigned long long m_leaf  = __ballot(alive && node < 0);
        const bool               inner_phase = __popcll(m_inner) >= kLeafBatch || m_leaf == 0ull;
        bool pop = false;
        if (inner_phase)
        {
            if (alive && node >= 0)
            {
                const float4 q0 = bvh.nodes[4 * node + 0], q1 = bvh.nodes[4 * node + 1], q2 = bvh.nodes[4 * node + 2],
                             q3 = bvh.nodes[4 * node + 3];
                float      tn0, tn1;
                const bool h0 = slab(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, best_t, tn0);
                const bool h1 = slab(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, best_t, tn1);
                const int  c0 = (int)f2u(q3.z), c1 = (int)f2u(q3.w);
                pop           = true;
                if (h0 && h1)
                {
                    const bool swap = tn1 < tn0;
                    if (sp < STACK) stack[(sp++) * kBlock] = (uint32_t)(swap ? c0 : c1);
                    node = swap ? c1 : c0;
                    pop  = false;
                }
                else if (h0 || h1)
                {
                    node = h0 ? c0 : c1;
                    pop  = false;
                }
            }
        }
        else if (alive && node < 0)
        {
            const uint32_t code = (uint32_t)~node, first = code & kLeafFirstMask, last = first + (code >> kLeafCountShift);
            for (uint32_t leaf = first; leaf <= last; ++leaf)
            {
                const float4 t0 = bvh.tris[4 * leaf + 0], t1 = bvh.tris[4 * leaf + 1], t2 = bvh.tris[4 * leaf + 2];
                float        t, u, v;
                if (tri_test(r, t0, t1, t2, t, u, v))
                {
                    const uint32_t gid = f2u(bvh.tris[4 * leaf + 3].x);
                    if (t < best_t || (t == best_t && gid < best_gid)) best_t = t, best_u = u, best_v = v, best_gid = gid;
                }
            }
            pop = true;
        }
        if (pop)
        {
            if (sp == 0)
            {
                hits[out] = make_float4(best_u, best_v, u2f(best_gid), best_t);
                alive     = false;
            }
            else
                node = (int)stack[(--sp) * kBlock];
        }
    }
}

// the wide traversal paths need every thread of the (1-D) grid to own a slice of the spill area
static BvhDev for_grid(const BvhDev& bvh, uint32_t grid_blocks)
{
    BvhDev b = bvh;
    if ((uint64_t)grid_blocks * kBlock > b.spill_threads) b.wide8_ok = 0;
    return b;
}

// The lane-refill kernels deal their chunks out statically (wave w takes slots w, w + W, ...): a workgroup that is not resident
// from the start runs its whole share after the others have finished.  Their grids are therefore clamped to what the runtime
// says fits at once (measured on the 262 k-triangle scene: 5 workgroups per CU requested with 32-KB stacks, 4 resident,
// closest hit 11.3 ms; 24-KB stacks, 5 resident, 7.9 ms).
template <auto K>
static uint32_t resident_grid(const LaunchCfg& cfg, uint32_t want)
{
    static int per_cu = -1;
    if (per_cu < 0)
    {
        int n = 0;
        per_cu = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, K, (int)kBlock, 0) == hipSuccess && n > 0) ? n : 0;
        if (cfg.sw_on(SW_TRACE_LAUNCHES)) fprintf(stderr, "[cap] resident workgroups per CU: %d\n", per_cu);
    }
    if (!cfg.cu_count || !per_cu) return want;
    const uint32_t cap = cfg.cu_count * (uint32_t)per_cu;
    return want < cap ? want : cap;
}

void launch_trace_primary(const LaunchCfg& cfg, const BvhDev& bvh, const CameraDev& cam, const ScreenDev& screen,
                          const FrameConst* frames, uint32_t n_slots, float4* hits, uint32_t* work)
{
    const uint32_t chunks = screen.pixels_padded >> 6;
    const bool no_packet = cfg.sw_on(SW_NO_PACKET);  // A/B switch
    if (cfg.stack_entries != 0 && work && bvh.tri_count >= 2 && !no_packet)
    {
        // issue-bound and light on registers: as many waves as fit
        uint32_t gx = (chunks * n_slots + 3) / 4;
        const uint32_t cap = cfg.cu_count ? resident_grid<k_trace_primary_packet<0>>(cfg, ~0u) : cfg.grid_blocks;
        if (gx > cap) gx = cap;
        if (gx == 0) gx = 1;
        hipLaunchKernelGGL(k_trace_primary_packet<0>, dim3(gx), dim3(kBlock), 0, cfg.stream, bvh, cam, screen, frames, n_slots, hits, work);
        return;
    }
    uint32_t       gx     = (chunks + 3) / 4;
    if (gx > cfg.grid_blocks) gx = cfg.grid_blocks;
    if (gx == 0) gx = 1;
    const dim3 grid(gx, n_slots);
    BvhDev     b = bvh;
    b.wide8_ok   = 0;  // 2-D grid: no per-thread spill slice
    if (cfg.stack_entries == 0)
        hipLaunchKernelGGL(k_trace_primary<0>, grid, dim3(kBlock), 0, cfg.stream, b, cam, screen, frames, hits);
    else if (cfg.stack_entries <= 32)
        hipLaunchKernelGGL(k_trace_primary<32>, grid, dim3(kBlock), 0, cfg.stream, b, cam, screen, frames, hits);
    else
        hipLaunchKernelGGL(k_trace_primary<64>, grid, dim3(kBlock), 0, cfg.stream, b, cam, screen, frames, hits);
}

static uint32_t queue_grid(const LaunchCfg& cfg, uint32_t max_count)
{
    uint32_t g = (max_count + kBlock - 1) / kBlock;
    if (g > cfg.grid_blocks) g = cfg.grid_blocks;
    return g ? g : 1;
}

void launch_trace_closest(const LaunchCfg& cfg, const BvhDev& bvh, const RayQueue& q, uint32_t max_count, float4* hits)
{
    dim3 grid(queue_grid(cfg, max_count));
    if (cfg.stack_entries == 0)
        hipLaunchKernelGGL(k_trace_closest<0>, grid, dim3(kBlock), 0, cfg.stream, bvh, q, hits);
    else if (cfg.stack_entries <= 32)
    {
        grid.x = resident_grid<k_trace_closest_refill<32>>(cfg, grid.x);
        hipLaunchKernelGGL(k_trace_closest_refill<32>, grid, dim3(kBlock), 0, cfg.stream, bvh, q, hits);
    }
    else
    {
        grid.x = resident_grid<k_trace_closest_refill<64>>(cfg, grid.x);
        hipLaunchKernelGGL(k_trace_closest_refill<64>, grid, dim3(kBlock), 0, cfg.stream, bvh, q, hits);
    }
}

void launch_trace_any(const LaunchCfg& cfg, const BvhDev& bvh, const ShadowQueue& q, uint32_t max_count, float4* target,
                      uint32_t pixels_padded, uint32_t n_slots, uint64_t* guard, uint32_t* work, bool mostly_unoccluded,
                      const FrameConst* frames)
{
    dim3         grid(queue_grid(cfg, max_count));
    const BvhDev bw = for_grid(bvh, grid.x);
    // shadow rays share one direction per frame and retire early: the plain per-chunk kernel beats the refill variant here
    // (8.3 vs 10.5 ms on the 262 k-triangle scene when it was tried)
    const uint32_t pre = (cfg.stack_entries == 0 && !mostly_unoccluded) ? pre_table_bytes(n_slots, bvh.fan_pair_count) : 0u;
    const bool     no_probe = cfg.sw_on(SW_NO_ANY_PROBE);  // A/B switch
    const uint32_t probe    = (uint32_t)cfg.sw_get(SW_ANY_PROBE, CAP_ANY_PROBE);
    if (pre != 0u && work && !no_probe && !cfg.any_no_probe && bvh.fan_pair_count <= kExhaustiveMax / 2)
    {
        // (3 .. 8 workgroups per CU measure the same: what is left is the planes' scattered read-modify-write traffic)
        const uint32_t per_cu = (uint32_t)cfg.sw_get(SW_ANY_BLOCKS, 6);
        uint32_t g = (max_count + kBlock - 1) / kBlock;
        const uint32_t cap = cfg.cu_count ? cfg.cu_count * per_cu : cfg.grid_blocks;
        g = g > cap ? cap : (g ? g : 1u);
        grid = dim3(g);
        hipLaunchKernelGGL(k_trace_any_small, grid, dim3(kBlock), pre, cfg.stream, bw, q, target, pixels_padded, n_slots, guard, work, frames, probe);
        return;
    }
#define CAP_LAUNCH_ANY(S, R) \
    hipLaunchKernelGGL((k_trace_any<S, R>), grid, dim3(kBlock), pre, cfg.stream, bw, q, target, pixels_padded, n_slots, guard, work, frames, pre)
    if (cfg.stack_entries == 0)
    {
        if (mostly_unoccluded) CAP_LAUNCH_ANY(0, true); else CAP_LAUNCH_ANY(0, false);
    }
    else if (bw.wide8_ok)
    {
        // wide traversal only (its stack continues in the spill slice): the smaller LDS part lets more workgroups be resident.
        // The binary code in this instantiation is never reached -- it has no spill and would drop entries past the LDS part.
        if (mostly_unoccluded) CAP_LAUNCH_ANY((int)kWideLdsEntries, true); else CAP_LAUNCH_ANY((int)kWideLdsEntries, false);
    }
    else if (cfg.stack_entries <= 32)
    {
        if (mostly_unoccluded) CAP_LAUNCH_ANY(32, true); else CAP_LAUNCH_ANY(32, false);
    }
    else
    {
        if (mostly_unoccluded) CAP_LAUNCH_ANY(64, true); else CAP_LAUNCH_ANY(64, false);
    }
#undef CAP_LAUNCH_ANY
}

// ------------------------------------------------------------------------------------------------
// Shading
// ------------------------------------------------------------------------------------------------
// sampling.h:13-23 with the texel pre-divided by 255 on the host (identical fp32 quotient).
__device__ __forceinline__ void bluenoise4x4(const float2* tex, uint32_t x, uint32_t y, uint32_t count, float& s0, float& s1)
{
    const uint32_t px = (count % 16u) % 4u, py = (count % 16u) / 4u;
    const uint32_t sx = (x * 4u + px) % 256u, sy = (y * 4u + py) % 256u;
    const float2   t  = tex[sy * 256u + sx];
    const float    k  = 0.61803398875f * (float)(count / 16u);
    const float    a = t.x + k, b = t.y + k;
    s0 = a - floorf(a);
    s1 = b - floorf(b);
}

// sampling.h:91-111
// sampling.h:91-111.  The two branches do the same arithmetic on (n.z, n.y) or (n.y, n.x): selecting the operands first keeps
// the values bit for bit and spares a wave with both kinds of normals (any wave in a box scene) one sqrt and two divisions.
__device__ __forceinline__ v3 ortho_vector(v3 n)
{
    const bool  zn = fabsf(n.z) > 0.0f;
    const float a = zn ? n.z : n.y, b = zn ? n.y : n.x;
    const float k = sqrtf(fmaf(a, a, b * b));
    const float q1 = a / k, q2 = b / k;
    return zn ? mk3(0.0f, -q1, q2) : mk3(q1, -q2, 0.0f);
}

// sampling.h:113-132 with e = 1 (shading.h:26): pow(1 - r2, 1/2) == sqrt(1 - r2)
__device__ __forceinline__ v3 map_to_hemisphere(float r1, float r2, v3 n)
{
    v3       u = ortho_vector(n);
    const v3 v = cross3(u, n);
    u          = cross3(n, v);
    float sin_psi, cos_psi;
    sincos_c((2.0f * kPi) * r1, sin_psi, cos_psi);
    const float cos_theta = sqrtf(1.0f - r2);
    const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    const float a = sin_theta * cos_psi, b = sin_theta * sin_psi;
    return normalize3(mk3(fmaf(n.x, cos_theta, fmaf(v.x, b, u.x * a)), fmaf(n.y, cos_theta, fmaf(v.y, b, u.y * a)),
                          fmaf(n.z, cos_theta, fmaf(v.z, b, u.z * a))));
}

// math_functions.h:36-47
__device__ __forceinline__ void oct_encode(v3 n, float& ox, float& oy)
{
    const float s = fabsf(n.x) + fabsf(n.y) + fabsf(n.z);
    n             = mk3(n.x / s, n.y / s, n.z / s);
    ox = n.x, oy = n.y;
    if (!(n.z >= 0.0f))
    {
        ox = (1.0f - fabsf(n.y)) * (n.x >= 0.0f ? 1.0f : -1.0f);
        oy = (1.0f - fabsf(n.x)) * (n.y >= 0.0f ? 1.0f : -1.0f);
    }
    ox = ox * 0.5f + 0.5f;
    oy = oy * 0.5f + 0.5f;
}

__device__ __forceinline__ uint32_t wrap_texel(float f, uint32_t n)
{
    const float m = f - floorf(f / (float)n) * (float)n;
    int         i = (int)m;
    if (i < 0) i = 0;
    if ((uint32_t)i >= n) i = 0;
    return (uint32_t)i;
}

// SampleLevel(..., 0) bilinear + WRAP on RGBA8 (scene.h:57, raytracing_system.cpp:377)
__device__ __forceinline__ v3 sample_texture(const TextureDev& tex, float u, float v)
{
    const float    fx = fmaf(u, (float)tex.width, -0.5f), fy = fmaf(v, (float)tex.height, -0.5f);
    const float    x0f = floorf(fx), y0f = floorf(fy);
    const float    wx = fx - x0f, wy = fy - y0f;
    const uint32_t x0 = wrap_texel(x0f, tex.width), y0 = wrap_texel(y0f, tex.height);
    // One 16-byte load: a texture is stored as the bilinear footprint of every texel -- (x, y), (x + 1, y), (x, y + 1), (x + 1, y + 1)
    // with WRAP applied, four RGBA8 words (cap_texture_upload).  Four scattered 4-byte loads per vertex were a quarter of the shade
    // stage's time on the textured scene (the stage is bound by the number of divergent addresses it sends);
    // the price is 4 x the texture memory.
    const uint4  fq   = reinterpret_cast<const uint4*>(tex.quads)[y0 * tex.width + x0];
    auto         rgba = [](uint32_t wd) { return make_uchar4((uint8_t)wd, (uint8_t)(wd >> 8), (uint8_t)(wd >> 16), (uint8_t)(wd >> 24)); };
    const uchar4 c00 = rgba(fq.x), c10 = rgba(fq.y), c01 = rgba(fq.z), c11 = rgba(fq.w);
    // byte / 255.0f without the division sequence: q = b * fl(1/255) is off by at most one ulp, and one residual step,
    // q + fl(b - 255 q) * fl(1/255), lands on the correctly rounded quotient for every one of the 256 bytes (checked exhaustively
    // in exact arithmetic: tests/test_oracle_kat.py::test_unorm8_is_the_division) -- 3 instructions instead of ~11
    auto unorm8 = [](uint8_t b) {
        const float r = 1.0f / 255.0f, fb = (float)b, q = fb * r;
        return fmaf(fmaf(-q, 255.0f, fb), r, q);
    };
    auto           lerp2 = [&](uint8_t a00, uint8_t a10, uint8_t a01, uint8_t a11) {
        const float f00 = unorm8(a00), f10 = unorm8(a10), f01 = unorm8(a01), f11 = unorm8(a11);
        const float top = fmaf(f10 - f00, wx, f00);
        const float bot = fmaf(f11 - f01, wx, f01);
        return fmaf(bot - top, wy, top);
    };
    return mk3(lerp2(c00.x, c10.x, c01.x, c11.x), lerp2(c00.y, c10.y, c01.y, c11.y), lerp2(c00.z, c10.z, c01.z, c11.z));
}

// Append one item per active lane to a device queue: one atomic per wave (64-lane ballot + popcount prefix).
__device__ __forceinline__ uint32_t wave_append(bool emit, uint32_t* counter)
{
    const unsigned long long mask = __ballot(emit);
    if (mask == 0ull) return 0;
    const uint32_t lane   = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t       base   = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// One path vertex: rt_direct_lighting.hlsl:38-83 (bounce 0) / one iteration of the rt_indirect.hlsl:91-174 loop, followed by
// the 64-lane compaction of the shadow ray and the extension ray into the class-`klass` sub-queues.  Called wave-uniformly
// (every lane of the wave, active or not) by the stand-alone shade kernel and by the fused trace+shade kernel.
//
// Everything the vertex needs that does not depend on the hit (pixel coordinates, the frame's light, the blue-noise sample) is
// fetched by shade_prefetch(); the fused kernel calls it BEFORE the triangle loop so that these dependent loads land under
// the loop's ALU work instead of in the latency-bound tail.
struct ShadePre
{
    bool  valid;
    v3    L, I;    // lighting.h:20-33 of this path's frame
    float r1, r2;  // sampling.h:13-23 sample of (pixel, frame * 25 + bounce)
    float r3, r4, r5, r6;  // EXT only: B, A of the same texel; R, G of the texel of count + 7
    float r1n, r2n;        // CARRY only: the sample of the path's NEXT vertex (count + 1), handed on in the queue entry
    bool  indirect_on;     // false for the three pixels of a 2x2 block that get no indirect sample this frame (LOWRES_INDIRECT)
};

// The per-frame constants of the batch (48 B x n_slots <= 3 KB) are staged in LDS once per workgroup: a path finds its frame's
// light and sample counter with a ~64-cycle ds_read instead of a global load that the blue-noise fetch would have to wait for.
__device__ __forceinline__ void stage_frames(const ShadeArgs& a, FrameConst* lds_frames)
{
    const uint32_t  words = a.n_slots * (uint32_t)(sizeof(FrameConst) / 4);
    const uint32_t* src   = reinterpret_cast<const uint32_t*>(a.frames);
    uint32_t*       dst   = reinterpret_cast<uint32_t*>(lds_frames);
    for (uint32_t i = threadIdx.x; i < words; i += kBlock) dst[i] = src[i];
    __syncthreads();
}

// CARRY (fused reference-model kernels): an extension ray's tmin / tmax are the constants kRayEps / kRayFar, so the two .w
// slots of its queue entry carry the blue-noise sample of the vertex it will find.  The vertex that emits the ray fetches
// that sample next to its other inputs, where nothing waits for it before the final stores; the vertex that receives it starts
// shading without a dependent global load.  carried_* = the .w slots of the entry this vertex came from (bounce >= 1).
template <bool EXT = false, bool FIRST = true, bool CARRY = false>
__device__ __forceinline__ ShadePre shade_prefetch(const ShadeArgs& a, const FrameConst* lds_frames, bool active, uint32_t pid,
                                                   float carried_r1 = 0.f, float carried_r2 = 0.f)
{
    ShadePre       s;
    const uint32_t slot = pid >> kPidShift, pl = pid & kPidMask;
    uint32_t       x = 0, y = 0;
    s.valid = active && local_pixel_to_xy(a.screen, pl, x, y);
    s.L = mk3(0, 0, 0), s.I = mk3(0, 0, 0), s.r1 = 0.f, s.r2 = 0.f;
    s.r3 = s.r4 = s.r5 = s.r6 = 0.f;
    s.r1n = s.r2n = 0.f;
    s.indirect_on = true;
    if (s.valid)
    {
        const FrameConst fc = lds_frames[slot < kMaxFrameSlots ? slot : 0];
        if (fc.lowres_sel & 4u) s.indirect_on = (x & 1u) == ((fc.lowres_sel >> 1) & 1u) && (y & 1u) == (fc.lowres_sel & 1u);
        s.L = mk3(fc.light_dir[0], fc.light_dir[1], fc.light_dir[2]);
        s.I = mk3(fc.light_intensity[0], fc.light_intensity[1], fc.light_intensity[2]);
        const uint32_t count = fc.frame_count * 25u + a.bounce;
        if (CARRY && !FIRST)
            s.r1 = carried_r1, s.r2 = carried_r2;
        else
            bluenoise4x4(a.scene.bluenoise, x, y, count, s.r1, s.r2);  // rt_indirect.hlsl:149
        if (CARRY) bluenoise4x4(a.scene.bluenoise, x, y, count + 1u, s.r1n, s.r2n);
        if (EXT)
        {
            bluenoise4x4(a.scene.bluenoise_ba, x, y, count, s.r3, s.r4);
            bluenoise4x4(a.scene.bluenoise, x, y, count + 7u, s.r5, s.r6);
        }
    }
    return s;
}

// Both queue appends of a wave with ONE device atomic: the extension and the shadow counter of a class sit in one 64-bit word
// (low half = extension entries, high half = shadow entries).
//
// Overflow guard (round 4).  A sub-queue's capacity is static because a path keeps the class it got at bounce 0 (cap_device.h); the
// appends used to rest on that argument alone, and anything that re-classifies paths -- the XCD-band experiment of round 3, any
// future sort -- would have written past the class's region, into its neighbour's entries or, for class 63, past the allocation.
// Now a lane whose slot lies beyond `capacity` does not store (emit_* comes back false for it), the wave that saw it bumps word 4
// of the guard block (CapStats::guard_append) and the consumers, which already clamp a class's count to its capacity, never read
// what was not written.  A run in which the guard fired has lost paths: bench.py and the tests treat it as a failure.
__device__ __forceinline__ void wave_append2(bool& emit_ext, bool& emit_shadow, uint32_t* counter_pair, uint32_t& ext_slot,
                                             uint32_t& shadow_slot, uint32_t capacity, uint64_t* guard)
{
    const unsigned long long me = __ballot(emit_ext), ms = __ballot(emit_shadow);
    ext_slot = shadow_slot = 0;
    if ((me | ms) == 0ull) return;
    const uint32_t lane   = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__ffsll((long long)(me | ms)) - 1u;
    uint32_t       lo = 0, hi = 0;
    if (lane == leader)
    {
        const unsigned long long add = ((unsigned long long)__popcll(ms) << 32) | (unsigned long long)__popcll(me);
        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(counter_pair), add);
        lo = (uint32_t)old, hi = (uint32_t)(old >> 32);
    }
    lo = __shfl(lo, (int)leader), hi = __shfl(hi, (int)leader);
    const unsigned long long below = (1ull << lane) - 1ull;
    ext_slot    = lo + (uint32_t)__popcll(me & below);
    shadow_slot = hi + (uint32_t)__popcll(ms & below);
    if (lo + (uint32_t)__popcll(me) > capacity || hi + (uint32_t)__popcll(ms) > capacity)  // wave-uniform, never true in a correct run
    {
        if (lane == leader)
        {
            atomicAdd((unsigned long long*)guard + 4, 1ull);
            guard[3] = ((uint64_t)lo << 32) | hi;
        }
        emit_ext    = emit_ext && ext_slot < capacity;
        emit_shadow = emit_shadow && shadow_slot < capacity;
    }
}

// Diagnostic build only (-DCAP_STAMPS): per-phase shader-clock sums of the fused kernel, see tools/stamps.py.
#ifdef CAP_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ unsigned long long g_wave_times[2 * 16384];  // (start, end) s_memrealtime of every wave of the LAST fused launch
struct Stamps
{
    unsigned long long last, acc[8], t_begin;
    __device__ void    start()
    {
        for (int i = 0; i < 8; ++i) acc[i] = 0;
        last    = __builtin_amdgcn_s_memtime();
        t_begin = __builtin_amdgcn_s_memrealtime();
    }
    __device__ void mark(int i, bool wait)
    {
#ifdef CAP_STAMPS_PHASES
        if (wait) __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        acc[i] += now - last;
        last = now;
#endif
    }
    __device__ void flush()
    {
        if ((threadIdx.x & 63u) == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(&g_stamps[i], acc[i]);
    }
};
#define STAMP(st, i, wait) (st).mark(i, wait)
#else
struct Stamps
{
    __device__ void start() {}
    __device__ void flush() {}
};
#define STAMP(st, i, wait) ((void)0)
#endif

// SKY_RMW: the sky term goes to the plane by load-add-store instead of three float atomics (the stand-alone shade stage)
// Probe (ShadeArgs::inline_probe, fused small-scene kernels): probe_rows = the PairPre rows of the probe pair per frame slot (two
// float4 each, LDS), probe_pairs = BvhDev::fan_pairs, probe_k the pair; n_probed counts the shadow rays the probe answered
// Ring (ShadeArgs::wave_ring): the survivors of the probe are not queued for another launch but parked in a ring of 128 entries
// that belongs to this wave alone (ring_org / ring_con: its slice of the shadow queue's memory); the kernel traces them 64 at a
// time itself (k_trace_shade trace_ring).  ring_head / ring_n are wave-uniform.
struct ProbeArgs
{
    const float4* rows  = nullptr;
    const float4* pairs = nullptr;
    uint32_t      k     = 0;
    float4*       ring_org = nullptr;
    float4*       ring_con = nullptr;
};
constexpr uint32_t kWaveRing = 128;  // <= 63 parked + <= 64 new
template <bool FIRST, bool FB = false, bool CARRY = false, bool SKY_RMW = false, bool PROBE = false>
__device__ __forceinline__ void shade_vertex(const ShadeArgs& a, const float4* shade_tab, const ShadePre& pre, uint32_t klass,
                                             uint32_t pid, float4 hit, v3 thr, uint32_t& n_shaded, Stamps& st,
                                             const ProbeArgs probe = ProbeArgs(), uint32_t* n_probed = nullptr, uint32_t ring_head = 0,
                                             uint32_t* ring_n = nullptr)
{
    const uint32_t Ppad = a.screen.pixels_padded;
    const uint32_t slot = pid >> kPidShift, pl = pid & kPidMask;
    {
        const size_t plane_idx = (size_t)slot * Ppad + pl;
        bool         valid = pre.valid;
        if (valid && (slot >= a.n_slots || pl >= Ppad))
        {
            // never true for a well-formed queue; reported through CapStats::guard_* instead of faulting
            atomicAdd((unsigned long long*)a.shaded_counter + 1, 1ull);
            a.shaded_counter[3] = ((uint64_t)a.bounce << 32) | pid;
            valid = false;
        }
        const uint32_t gid = f2u(hit.z);

        bool   emit_shadow = false, emit_ext = false;
        v3     p = mk3(0, 0, 0), dir = mk3(0, 0, 0), contrib = mk3(0, 0, 0);

        if (FIRST && !valid)
        {
            // padding lane of a partial / absent tile: define the planes so the resolve adds exact zeros
            a.planes.color[plane_idx]  = make_float4(0, 0, 0, 0);
            a.planes.direct[plane_idx] = make_float4(0, 0, 0, 0);  // albedo_in_w: code 0
            if (!a.albedo_in_w) a.planes.albedo[plane_idx] = make_float4(0, 0, 0, 0);
        }
        if (valid && gid == kInvalidId)
        {
            if (FIRST)
            {
                // rt_direct_lighting.hlsl:53-59, rt_indirect.hlsl:75-79
                a.planes.color[plane_idx]  = make_float4(0.f, 0.f, 0.f, 1.f);
                a.planes.direct[plane_idx] = make_float4(0.7f, 0.7f, 0.85f, 1.f);  // albedo_in_w: code 1
                if (!a.albedo_in_w) a.planes.albedo[plane_idx] = make_float4(1.f, 1.f, 1.f, 1.f);
                if (slot == a.aov_slot) a.planes.aov_normal_depth[pl] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            else
            {
                // rt_indirect.hlsl:94-99  color += throughput * sky.  Exactly one lane in the whole grid owns this path, so the
                // three no-return float atomics are plain IEEE adds in program order; unlike a load-add-store they do not make
                // the wave wait for the old value.
                // (A load-add-store here instead, as in k_trace_any: 16.4 -> 16.6 ms.)
                if (SKY_RMW)
                {
                    // the tree path's shade stage: three scattered float atomics per escaping ray are three 64-B memory-side
                    // requests each; the path is this entry's only writer within the launch, so a 16-B load-add-store gives the
                    // same IEEE additions
                    const float4 cur = a.planes.color[plane_idx];
                    a.planes.color[plane_idx] = make_float4(cur.x + thr.x * 0.7f, cur.y + thr.y * 0.7f, cur.z + thr.z * 0.85f, cur.w);
                }
                else
                {
                    float* c = reinterpret_cast<float*>(a.planes.color + plane_idx);
                    atomicAdd(c + 0, thr.x * 0.7f);
                    atomicAdd(c + 1, thr.y * 0.7f);
                    atomicAdd(c + 2, thr.z * 0.85f);
                }
            }
        }
        else if (valid)
        {
            ++n_shaded;
            // scene.h:5-50 InterpolateAttributes on the pre-gathered triangle record
            const float4* tab = shade_tab + kShadeRec * (size_t)gid;
            const float4  s0 = tab[0], s1 = tab[1], s2 = tab[2], s3 = tab[3], s4 = tab[4], s5 = tab[5];
            const float   u = hit.x, v = hit.y, w = (1.0f - u) - v;
            auto          mix = [&](float c0, float c1, float c2) { return fmaf(c2, v, fmaf(c1, u, c0 * w)); };
            const v3      n = normalize3(mk3(mix(s3.x, s4.x, s5.x), mix(s3.y, s4.y, s5.y), mix(s3.z, s4.z, s5.z)));
            p = mk3(mix(s0.x, s1.x, s2.x), mix(s0.y, s1.y, s2.y), mix(s0.z, s1.z, s2.z));
            // scene.h:52-61 GetMaterial
            v3       kd   = mk3(a.scene.kd_untextured, a.scene.kd_untextured, a.scene.kd_untextured);
            uint32_t inst = 0;
            if (a.scene.texture_count != 0 || (FIRST && slot == a.aov_slot))  // wave-uniform: untextured scenes skip the dependent load
            {
                const float4 idf   = tab[6];  // (instance, primitive, mesh_texture[instance]) in the record itself: no second fetch
                inst               = f2u(idf.x);
                const uint32_t tex = f2u(idf.z);
                if (tex != kInvalidId && tex < a.scene.texture_count)
                {
                    const float tu = mix(s0.w, s2.w, s4.w), tv = mix(s1.w, s3.w, s5.w);
                    const v3    c  = sample_texture(a.scene.textures[tex], tu, 1.0f - tv);
                    kd             = mk3(pow22_c(c.x), pow22_c(c.y), pow22_c(c.z));
                }
            }
            const bool black = kd.x < 1e-5f && kd.y < 1e-5f && kd.z < 1e-5f;  // rt_direct_lighting.hlsl:68, rt_indirect.hlsl:108
            if (FIRST)
            {
                a.planes.color[plane_idx]  = make_float4(0.f, 0.f, 0.f, 1.f);
                // albedo_in_w (untextured scene, accumulate-only render): the albedo is one of four constants, so its plane is
                // neither written nor read; direct.w carries which -- 0: (0,0,0) padding, 1: (1,1,1) sky, 2: the untextured kd, 3: black
                a.planes.direct[plane_idx] = make_float4(0.f, 0.f, 0.f, a.albedo_in_w ? (black ? 3.f : 2.f) : 1.f);
                if (!a.albedo_in_w) a.planes.albedo[plane_idx] = black ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(kd.x, kd.y, kd.z, 1.f);
                if (slot == a.aov_slot)
                {
                    float4 nd = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!black)
                    {
                        oct_encode(n, nd.x, nd.y);
                        nd.z = (float)inst;
                        nd.w = length3(mk3(a.cam.position[0], a.cam.position[1], a.cam.position[2]) - p);
                    }
                    a.planes.aov_normal_depth[pl] = nd;
                }
            }
            bool reused = false;
            if (FB && !FIRST && !black)
            {
                // rt_indirect.hlsl:116-145 GBUFFER_FEEDBACK: a vertex the previous frame saw (inside its image, depth within 5 %)
                // takes that frame's shaded, TAA'd colour and ends the path.  A NaN uv counts as disocclusion (stated choice:
                // HLSL's any(uv < 0) || any(uv > 1) would let it through to an undefined texel address).
                const uint32_t W = a.screen.width, H = a.screen.height;
                const f2       puv = image_plane_uv(a.fb.prev_cam, p);
                if (puv.x >= 0.0f && puv.y >= 0.0f && puv.x <= 1.0f && puv.y <= 1.0f)
                {
                    const f2    pxy        = uv_to_xy(puv, W, H);
                    const float prev_depth = ldi(Img{a.fb.prev_normal_depth, W, H}, (int)pxy.x, (int)pxy.y).w;
                    const float cur_depth =
                        length3(p - mk3(a.fb.prev_cam.position[0], a.fb.prev_cam.position[1], a.fb.prev_cam.position[2]));
                    if (!(fabsf(prev_depth - cur_depth) / cur_depth > 0.05f))
                    {
                        reused        = true;
                        const v3 hc   = sample_bilinear(Img{a.fb.color_history, W, H}, puv);
                        // the path is this entry's only writer within the launch (it either escapes or is shaded), and bounce 0 defined the
                        // entry one launch ago: a 16-B load-add-store makes the same IEEE additions as three float atomics.  Unlike the sky
                        // term above, where the atomics win by 1 %, here most vertices of a frame take this branch -- the previous frame saw
                        // them -- and 6 M atomics per launch cost more than the load's latency: real-time frame 0.648 -> 0.638 ms.
                        const float4 cur = a.planes.color[plane_idx];
                        a.planes.color[plane_idx] = make_float4(cur.x + thr.x * hc.x, cur.y + thr.y * hc.y, cur.z + thr.z * hc.z, cur.w);
                    }
                }
            }
            if (!black && !reused)
            {
                // lighting.h:35-61: unshadowed direct term; the visibility ray is queued for the any-hit kernel
                const float ndl = fmaxf(0.0f, dot3(n, pre.L));
                v3          c   = mk3(((pre.I.x * kd.x) * kInvPi) * ndl, ((pre.I.y * kd.y) * kInvPi) * ndl, ((pre.I.z * kd.z) * kInvPi) * ndl);
                if (c.x != 0.0f || c.y != 0.0f || c.z != 0.0f)
                {
                    emit_shadow = true;
                    contrib     = FIRST ? c : thr * c;  // rt_direct_lighting.hlsl:77 / rt_indirect.hlsl:136
                }
                // rt_indirect.hlsl:149-170
                dir             = map_to_hemisphere(pre.r1, pre.r2, n);
                const float ndd = dot3(n, dir);
                const float pdf = fmaxf(0.0f, ndd) / kPi;  // shading.h:19-22
                if (!(pdf < 1e-5f))
                {
                    const float f = (kInvPi * fmaxf(ndd, 0.0f)) / pdf;
                    thr           = thr * f;
                    if (!FIRST) thr = thr * kd;
                    // the reference traces one more ray after the last bounce whose payload is never read (:91,:173)
                    emit_ext = a.bounce < a.num_bounces && (!FIRST || pre.indirect_on);
                }
            }
        }

        if (PROBE)
        {
            if (probe.rows != nullptr)  // wave-uniform
            {
                // lanes without a shadow ray test an empty interval's worth of nothing: their result is discarded
                const Ray    sr  = make_ray(p, pre.L, kRayEps, kRayFar);
                const uint32_t ps = valid ? slot : 0u;
                const float4   pa = probe.rows[2u * ps], pb = probe.rows[2u * ps + 1u];
                const bool   occluded = pair_occludes_pre(sr, probe.pairs, probe.k, pa, pb);
                if (emit_shadow && occluded) emit_shadow = false, ++*n_probed;
            }
        }
        // a.shadow.count == a.out.count + 1: both counters of a class share one 64-bit word (one atomic per wave for both queues)
        uint32_t ei, si;
        STAMP(st, 2, true);  // shading inputs arrived + shading ALU
        wave_append2(emit_ext, emit_shadow, a.out.count + klass * kCounterStride, ei, si, a.out.class_capacity, a.shaded_counter);
        STAMP(st, 3, true);  // append atomic returned
        ei += klass * a.out.class_capacity;
        si += klass * a.shadow.class_capacity;
        if (PROBE && probe.ring_org != nullptr)  // wave-uniform
        {
            // the entry goes to this wave's own ring (the counter above still counted it: CapStats::shadow_entries)
            const unsigned long long ms = __ballot(emit_shadow);
            if (emit_shadow)
            {
                const uint32_t lane_ = threadIdx.x & 63u;
                const uint32_t pos   = (ring_head + *ring_n + (uint32_t)__popcll(ms & ((1ull << lane_) - 1ull))) & (kWaveRing - 1u);
                probe.ring_org[pos]  = make_float4(p.x, p.y, p.z, u2f(pid));
                probe.ring_con[pos]  = make_float4(contrib.x, contrib.y, contrib.z, 0.0f);
            }
            *ring_n += (uint32_t)__popcll(ms);
            // the entries are read by OTHER lanes of this wave (trace_ring): wavefront-scope release here, acquire there.  No code on
            // gfx950 (same-wave LDS and vector-memory operations retire in issue order), but it is what forbids the compiler to move
            // the stores below the loads -- ordering that until round 3 rested on a scheduling barrier alone.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        else if (emit_shadow)
        {
            // reference model: the shadow ray's direction is its frame's light and tmin / tmax are constants (lighting.h:39-47), so
            // the entry is 32 B -- (origin, path id) and the contribution; the any-hit kernel looks the direction up by frame slot
            a.shadow.org_tmin[si]    = make_float4(p.x, p.y, p.z, u2f(pid));
            a.shadow.contrib_pid[si] = make_float4(contrib.x, contrib.y, contrib.z, 0.0f);
        }
        if (emit_ext)
        {
            a.out.org_tmin[ei] = make_float4(p.x, p.y, p.z, CARRY ? pre.r1n : kRayEps);
            a.out.dir_tmax[ei] = make_float4(dir.x, dir.y, dir.z, CARRY ? pre.r2n : kRayFar);
            a.out.thr_pid[ei]  = make_float4(thr.x, thr.y, thr.z, u2f(pid));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// EXT shading model (no reference counterpart, specification in DESIGN.md "EXT shading model"):
// Lambert + GGX microfacet BSDF, emissive triangles sampled by area with one shadow ray per vertex (next-event estimation),
// emission seen directly only from the camera, black environment.  Same queues and kernels as the reference model.
// ------------------------------------------------------------------------------------------------
struct ExtBsdf
{
    v3    f;
    float pdf_spec, pdf_diff;
};

__device__ __forceinline__ float lum3(v3 c) { return fmaf(c.z, 0.114f, fmaf(c.y, 0.587f, c.x * 0.299f)); }

// What the BSDF needs of the vertex and the outgoing direction alone: both evaluations of a vertex (towards the light sample and
// along the sampled direction) share them, so they are computed once -- the same expressions on the same operands as before.
struct ExtView
{
    float cos_o, lam_o;  // lam_o = cos_o + sqrt(a2 + (1 - a2) cos_o^2): the outgoing direction's factor of the masking term
};
__device__ __forceinline__ ExtView ext_view(float a2, v3 nf, v3 wo)
{
    ExtView w;
    w.cos_o = dot3(nf, wo);
    w.lam_o = w.cos_o + sqrtf(fmaf(1.0f - a2, w.cos_o * w.cos_o, a2));
    return w;
}
__device__ __forceinline__ ExtBsdf ext_bsdf(v3 kd, v3 ks, float a2, v3 nf, v3 wo, v3 wi, const ExtView& vw)
{
    const float cos_i = dot3(nf, wi);
    const v3    h     = normalize3(wo + wi);
    const float cos_h = dot3(nf, h), woh = dot3(wo, h);
    const float dd    = fmaf(cos_h * cos_h, a2 - 1.0f, 1.0f);
    // D G / (4 cos_o cos_i) in its cancelled ("visibility") form, one division: see oracle/cap_oracle.cpp ext_bsdf (the same operations)
    const float pdd   = kPi * dd * dd;
    const float lam_i = cos_i + sqrtf(fmaf(1.0f - a2, cos_i * cos_i, a2));
#if defined(CAP_EXT_DIAG) && CAP_EXT_DIAG == 3  // diagnostic build: no microfacet term (D, G and their divisions fall away)
    const float spec  = 0.0f * (vw.cos_o + woh);
#else
    const float spec  = a2 / (pdd * (vw.lam_o * lam_i));
#endif
    ExtBsdf     r;
    r.f        = mk3(kd.x * kInvPi + ks.x * spec, kd.y * kInvPi + ks.y * spec, kd.z * kInvPi + ks.z * spec);
    r.pdf_spec = (a2 * cos_h) / (pdd * (4.0f * woh));
    r.pdf_diff = cos_i * kInvPi;
    return r;
}

// Tables of the EXT model in LDS (fused small-scene kernels, scenes of at most kExhaustiveMax triangles): the per-mesh materials,
// the light table and, per emissive triangle, its three vertices, unit normal and emission -- what shade_vertex_ext otherwise
// fetches through four dependent global loads and recomputes per vertex (the normal: a cross product and a normalisation that
// depend on the light triangle alone).  Same operations on the same operands, done once per workgroup.
struct ExtTables
{
    const MaterialDev* materials  = nullptr;  // [mesh]
    const float*       light_cdf  = nullptr;
    const float4*      light_rec  = nullptr;  // 4 per light: (q0, ke.x) (q1, ke.y) (q2, ke.z) (nl, -)
};
constexpr uint32_t kExtLightsMax = 32;  // lights the LDS table holds (more: the global path)

// INLINE (ShadeArgs::inline_nee, fused small-scene kernels only): bvh is traced for the shadow ray here; acc = what the path has
// gathered so far
template <bool FIRST, bool INLINE = false>
__device__ __forceinline__ void shade_vertex_ext(const ShadeArgs& a, const float4* shade_tab, const ShadePre& pre, uint32_t klass,
                                                 uint32_t pid, float4 hit, v3 thr,
                                                 v3 d, uint32_t& n_shaded, const BvhDev* bvh = nullptr, v3 acc = mk3(0.f, 0.f, 0.f),
                                                 const ExtTables tabs = ExtTables())
{
    const uint32_t Ppad = a.screen.pixels_padded;
    const uint32_t slot = pid >> kPidShift, pl = pid & kPidMask;
    const size_t   plane_idx = (size_t)slot * Ppad + pl;
    bool           valid = pre.valid;
    if (valid && (slot >= a.n_slots || pl >= Ppad))
    {
        atomicAdd((unsigned long long*)a.shaded_counter + 1, 1ull);
        a.shaded_counter[3] = ((uint64_t)a.bounce << 32) | pid;
        valid = false;
    }
    const uint32_t gid = f2u(hit.z);
    bool  emit_shadow = false, emit_ext = false;
    v3    p = mk3(0, 0, 0), dir = mk3(0, 0, 0), contrib = mk3(0, 0, 0), sdir = mk3(0, 0, 1), first_ke = mk3(0, 0, 0);
    float stmax = 0.0f;

    if (FIRST && !valid)
    {
        a.planes.color[plane_idx]  = make_float4(0, 0, 0, 0);
        a.planes.direct[plane_idx] = make_float4(0, 0, 0, 0);
        if (!a.albedo_in_w) a.planes.albedo[plane_idx] = make_float4(0, 0, 0, 0);
    }
    if (valid && gid == kInvalidId)
    {
        if (FIRST)
        {
            // black environment: the camera ray that leaves the scene carries nothing
            a.planes.color[plane_idx]  = make_float4(0.f, 0.f, 0.f, 1.f);
            a.planes.direct[plane_idx] = make_float4(0.f, 0.f, 0.f, 1.f);
            if (!a.albedo_in_w) a.planes.albedo[plane_idx] = make_float4(1.f, 1.f, 1.f, 1.f);  // else: direct.w == 1 says so
            if (slot == a.aov_slot) a.planes.aov_normal_depth[pl] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    else if (valid)
    {
        ++n_shaded;
        const float4* st = shade_tab + kShadeRec * (size_t)gid;
        const float4  s0 = st[0], s1 = st[1], s2 = st[2], s3 = st[3], s4 = st[4], s5 = st[5];
        const float   u = hit.x, v = hit.y, w = (1.0f - u) - v;
        auto          mix = [&](float c0, float c1, float c2) { return fmaf(c2, v, fmaf(c1, u, c0 * w)); };
        const v3      n = normalize3(mk3(mix(s3.x, s4.x, s5.x), mix(s3.y, s4.y, s5.y), mix(s3.z, s4.z, s5.z)));
        p = mk3(mix(s0.x, s1.x, s2.x), mix(s0.y, s1.y, s2.y), mix(s0.z, s1.z, s2.z));
        const uint32_t    inst = f2u(st[6].x);
        const MaterialDev m    = tabs.materials ? tabs.materials[inst] : a.scene.materials[inst];
        const v3    kd = mk3(m.kd[0], m.kd[1], m.kd[2]), ks = mk3(m.ks[0], m.ks[1], m.ks[2]), ke = mk3(m.ke[0], m.ke[1], m.ke[2]);
        const float alpha = fmaxf(m.roughness * m.roughness, 1e-3f), a2 = alpha * alpha;
        const v3    wo = mk3(-d.x, -d.y, -d.z);
        const v3    nf = dot3(n, wo) < 0.0f ? mk3(-n.x, -n.y, -n.z) : n;
        const ExtView vw = ext_view(a2, nf, wo);
        if (FIRST)
        {
            first_ke = ke;
            if (!INLINE)  // (INLINE: both written below, once the shadow ray is known / the path ends)
            {
                a.planes.color[plane_idx]  = make_float4(0.f, 0.f, 0.f, 1.f);
                a.planes.direct[plane_idx] = make_float4(ke.x, ke.y, ke.z, 1.f);
            }
            if (!a.albedo_in_w) a.planes.albedo[plane_idx] = make_float4(1.f, 1.f, 1.f, 1.f);  // this model folds kd into the throughput
            if (slot == a.aov_slot)
            {
                float4 nd;
                oct_encode(n, nd.x, nd.y);
                nd.z = (float)inst;
                nd.w = length3(mk3(a.cam.position[0], a.cam.position[1], a.cam.position[2]) - p);
                a.planes.aov_normal_depth[pl] = nd;
            }
        }
        // ---- next-event estimation: one point on the emissive triangles, uniform by area ----
#if defined(CAP_EXT_DIAG) && CAP_EXT_DIAG == 2  // diagnostic build: no next-event estimation at all
        if (false)
#else
        if (a.scene.light_count != 0)
#endif
        {
            const float target = pre.r4 * a.scene.light_area;
            const float* cdf   = tabs.light_cdf ? tabs.light_cdf : a.scene.light_cdf;
            uint32_t    lo = 0, hi = a.scene.light_count - 1;
            while (lo < hi)  // first entry whose prefix sum exceeds target, else the last
            {
                const uint32_t mid = (lo + hi) >> 1;
                if (cdf[mid] > target) hi = mid; else lo = mid + 1;
            }
            v3 q0, q1, q2, nl, lke;
            if (tabs.light_rec)  // wave-uniform
            {
                const float4 r0 = tabs.light_rec[4 * lo], r1 = tabs.light_rec[4 * lo + 1], r2 = tabs.light_rec[4 * lo + 2], r3 = tabs.light_rec[4 * lo + 3];
                q0 = mk3(r0.x, r0.y, r0.z), q1 = mk3(r1.x, r1.y, r1.z), q2 = mk3(r2.x, r2.y, r2.z), nl = mk3(r3.x, r3.y, r3.z);
                lke = mk3(r0.w, r1.w, r2.w);
            }
            else
            {
                const uint32_t lg = a.scene.light_tris[lo];
                const float4*  lt = shade_tab + kShadeRec * (size_t)lg;
                const float4   l0 = lt[0], l1 = lt[1], l2 = lt[2];
                q0 = mk3(l0.x, l0.y, l0.z), q1 = mk3(l1.x, l1.y, l1.z), q2 = mk3(l2.x, l2.y, l2.z);
                nl = normalize3(cross3(q1 - q0, q2 - q0));
                const MaterialDev lm = a.scene.materials[f2u(lt[6].x)];
                lke = mk3(lm.ke[0], lm.ke[1], lm.ke[2]);
            }
            const float    su = sqrtf(pre.r5), b0 = 1.0f - su, b1 = su * (1.0f - pre.r6), b2 = su * pre.r6;
            const v3 lp = mk3(fmaf(q2.x, b2, fmaf(q1.x, b1, q0.x * b0)), fmaf(q2.y, b2, fmaf(q1.y, b1, q0.y * b0)),
                              fmaf(q2.z, b2, fmaf(q1.z, b1, q0.z * b0)));
            const v3    Lv = lp - p;
            const float d2 = dot3(Lv, Lv), dist = sqrtf(d2);
            const v3    wi = Lv * (1.0f / dist);
            const float cos_s = dot3(nf, wi), cos_l = fabsf(dot3(nl, wi));
            if (cos_s > 0.0f && cos_l > 0.0f && d2 > 0.0f)
            {
                const ExtBsdf     bs  = ext_bsdf(kd, ks, a2, nf, wo, wi, vw);
                const float       wgt = ((cos_s * cos_l) * a.scene.light_area) / d2;
                const v3 c = mk3((thr.x * bs.f.x) * (lke.x * wgt), (thr.y * bs.f.y) * (lke.y * wgt), (thr.z * bs.f.z) * (lke.z * wgt));
                if (c.x != 0.0f || c.y != 0.0f || c.z != 0.0f)
                {
                    emit_shadow = true, contrib = c, sdir = wi, stmax = dist * 0.999f;
                }
            }
        }
        // ---- BSDF sampling: GGX half vector or cosine hemisphere, chosen by luminance ----
        const float ls = lum3(ks), sum = lum3(kd) + ls;
        if (sum > 0.0f)
        {
            const float ps = ls / sum;
            // The two lobes sample a polar angle -- GGX: cos^2 = (1 - r2) / (1 + (a2 - 1) r2) for the half vector, Lambert:
            // cos = sqrt(1 - r2) for the direction (MapToHemisphere, sampling.h:113-132, e = 1) -- around the SAME frame with
            // the SAME azimuth; a wave whose lanes chose different lobes (any wave: the choice is a random number per lane) used
            // to run the frame, the sincos and the normalisation twice.  One copy now, the polar angle selected per lane: the
            // same operations on the same operands for either lobe, so the same bits.
            const bool  lobe_spec = pre.r3 < ps;
            const float c2  = (1.0f - pre.r2) / fmaf(a2 - 1.0f, pre.r2, 1.0f);
            const float ctd = sqrtf(1.0f - pre.r2);
            const float ct  = lobe_spec ? sqrtf(c2) : ctd;
            const float stt = lobe_spec ? sqrtf(fmaxf(0.0f, 1.0f - c2)) : sqrtf(1.0f - ctd * ctd);
            float       sp, cp;
            sincos_c((2.0f * kPi) * pre.r1, sp, cp);
            v3       uu = ortho_vector(nf);
            const v3 vv = cross3(uu, nf);
            uu          = cross3(nf, vv);
            const float ca = stt * cp, cb = stt * sp;
            const v3    hh = normalize3(mk3(fmaf(nf.x, ct, fmaf(vv.x, cb, uu.x * ca)), fmaf(nf.y, ct, fmaf(vv.y, cb, uu.y * ca)),
                                            fmaf(nf.z, ct, fmaf(vv.z, cb, uu.z * ca))));
            const float k2 = 2.0f * dot3(wo, hh);
            const v3    wi = lobe_spec ? mk3(fmaf(hh.x, k2, -wo.x), fmaf(hh.y, k2, -wo.y), fmaf(hh.z, k2, -wo.z)) : hh;
            const float cos_i = dot3(nf, wi);
            if (cos_i > 0.0f)
            {
                const ExtBsdf bs  = ext_bsdf(kd, ks, a2, nf, wo, wi, vw);
                const float   pdf = ps * bs.pdf_spec + (1.0f - ps) * bs.pdf_diff;
                if (pdf > 1e-8f)
                {
                    const float wgt = cos_i / pdf;
                    thr      = mk3(thr.x * (bs.f.x * wgt), thr.y * (bs.f.y * wgt), thr.z * (bs.f.z * wgt));
                    dir      = wi;
                    emit_ext = a.bounce < a.num_bounces;
                }
            }
        }
    }
    uint32_t ei, si;
    wave_append2(emit_ext, emit_shadow, a.out.count + klass * kCounterStride, ei, si, a.out.class_capacity, a.shaded_counter);  // (INLINE: the shadow counter still counts the rays)
    ei += klass * a.out.class_capacity;
    si += klass * a.shadow.class_capacity;
    if (INLINE)
    {
        // the any-hit kernel's test and its addition, here: lanes without a shadow ray trace an empty interval
        const Ray  sr      = make_ray(p, sdir, kRayEps, emit_shadow ? stmax : kRayEps);
        bool       visible = false;
#if defined(CAP_EXT_DIAG) && CAP_EXT_DIAG == 1  // diagnostic build (wrong images, right timing): what the inline any-test costs
        visible = emit_shadow;
#else
#if defined(CAP_NEE_CHECK)  // diagnostic build: both lists, every disagreement counted (CapStats::guard_shade stays 0 when the rule holds)
        if (__ballot(emit_shadow) != 0ull)
        {
            const bool full = exhaustive_any<false>(*bvh, sr), part = exhaustive_any<false, true>(*bvh, sr);
            if (emit_shadow && full != part) atomicAdd((unsigned long long*)a.shaded_counter + 1, 1ull);
            visible = emit_shadow && !part;
        }
#else
        if (__ballot(emit_shadow) != 0ull) visible = emit_shadow && !exhaustive_any<false, true>(*bvh, sr);  // (a wave without a shadow ray: no test)
#endif
#endif
        const bool on_surface = valid && gid != kInvalidId;
        if (FIRST)
        {
            if (on_surface)
                a.planes.direct[plane_idx] = visible ? make_float4(first_ke.x + contrib.x, first_ke.y + contrib.y, first_ke.z + contrib.z, 1.f)
                                                     : make_float4(first_ke.x, first_ke.y, first_ke.z, 1.f);
        }
        else if (visible)
            acc = mk3(acc.x + contrib.x, acc.y + contrib.y, acc.z + contrib.z);
        // the path ends here unless it continues: its colour-plane entry is written exactly once (a camera ray that left the scene
        // wrote it above)
        if (valid && !emit_ext && (!FIRST || on_surface)) a.planes.color[plane_idx] = make_float4(acc.x, acc.y, acc.z, 1.f);
        if (emit_ext) a.out.acc[ei] = make_float4(acc.x, acc.y, acc.z, 0.f);
    }
    else if (emit_shadow)
    {
        a.shadow.org_tmin[si]    = make_float4(p.x, p.y, p.z, kRayEps);
        a.shadow.dir_tmax[si]    = make_float4(sdir.x, sdir.y, sdir.z, stmax);
        a.shadow.contrib_pid[si] = make_float4(contrib.x, contrib.y, contrib.z, u2f(pid));
    }
    if (emit_ext)
    {
        a.out.org_tmin[ei] = make_float4(p.x, p.y, p.z, kRayEps);
        a.out.dir_tmax[ei] = make_float4(dir.x, dir.y, dir.z, kRayFar);
        a.out.thr_pid[ei]  = make_float4(thr.x, thr.y, thr.z, u2f(pid));
    }
}

// Statistics of a launch: shaded vertices and (small-scene path) the shadow rays the producer's probe answered -- ONE 64-bit atomic
// per wave into words 2 (probed) and 3 (shaded) of the counter line of the wave's queue class, whose words 0 and 1 are the
// bounce's extension / shadow queue lengths (context.hip reads all four from the batch's counter copy).
// Until round 3 every wave of the grid added to ONE word at the end of every launch (and a second one for the probe count): a
// device-scope atomic on one address retires ~88 per microsecond (MI355X_MICROARCH.md "dequeue"), so the 6144 waves of a launch
// with little work -- which all finish together -- queued for 70 us behind each other: the whole "fixed cost" of the persistent
// launches that round 2's batch-size sweep measured (tools/tiny_trace.sh: 77 us per fused launch whatever its work, 6 us for
// the any-hit kernel, which has no such flush), and 17 % of a rank's step at eight shards.
__device__ __forceinline__ void flush_stats(uint32_t* class_line, uint32_t n_shaded, uint32_t n_probed = 0)
{
    for (int off = 32; off > 0; off >>= 1) n_shaded += __shfl_down(n_shaded, off), n_probed += __shfl_down(n_probed, off);
    if ((threadIdx.x & 63u) == 0 && (n_shaded | n_probed))
        atomicAdd(reinterpret_cast<unsigned long long*>(class_line + 2), ((unsigned long long)n_shaded << 32) | (unsigned long long)n_probed);
}

// Stand-alone shade stage (used with the LBVH stack traversal): consumes the hit records of the preceding trace kernel.
template <bool FIRST, bool EXT, bool FB = false>
__global__ __launch_bounds__(kBlock) void k_shade(ShadeArgs a)
{
    const uint32_t Ppad = a.screen.pixels_padded;
    // FIRST: identity queue, item i of frame slot blockIdx.y is local pixel i.  Otherwise: chunk slots of the input queue.
    const uint32_t chunks   = FIRST ? (Ppad >> 6) : (a.in.class_capacity >> 6) * kQueueClasses;
    uint32_t       n_shaded = 0;
    __shared__ FrameConst lds_frames[kMaxFrameSlots];
    stage_frames(a, lds_frames);
    Stamps st;
    st.start();
    for (uint32_t chunk = wave_global_id(); chunk < chunks; chunk += wave_total())
    {
        uint32_t i, klass;
        bool     active;
        if (FIRST)
        {
            i      = chunk * 64 + (threadIdx.x & 63u);
            active = true;
            klass  = chunk_class(blockIdx.y * (Ppad >> 6) + chunk);  // the path's class for its whole life
        }
        else
            active = queue_chunk(a.in.count, a.in.class_capacity, chunk, threadIdx.x & 63u, i, klass);
        uint32_t pid = 0;
        float4   hit = make_float4(0.f, 0.f, u2f(kInvalidId), 0.f);
        v3       thr = mk3(1.0f, 1.0f, 1.0f);
        if (active)
        {
            if (FIRST)
            {
                pid = (blockIdx.y << kPidShift) | i;
                hit = a.hits[(size_t)blockIdx.y * Ppad + i];
            }
            else
            {
                const float4 tp = a.in.thr_pid[i];
                thr = mk3(tp.x, tp.y, tp.z), pid = f2u(tp.w);
                hit = a.hits[i];
            }
        }
        const ShadePre pre = shade_prefetch<EXT>(a, lds_frames, active, pid);
        if constexpr (EXT)
        {
            // the EXT BSDF depends on the incoming direction: the camera ray (bounce 0) or the queue entry's direction
            v3 d = mk3(0.f, 0.f, 1.f);
            if (active)
            {
                if (FIRST)
                {
                    uint32_t x, y;
                    if (local_pixel_to_xy(a.screen, i, x, y)) d = primary_dir(a.cam, a.screen, a.frames[blockIdx.y], x, y);
                }
                else
                {
                    const float4 dq = a.in.dir_tmax[i];
                    d               = mk3(dq.x, dq.y, dq.z);
                }
            }
            shade_vertex_ext<FIRST>(a, a.scene.shade_tris, pre, klass, pid, hit, thr, d, n_shaded);
        }
        else
            shade_vertex<FIRST, FB, false, true>(a, a.scene.shade_tris, pre, klass, pid, hit, thr, n_shaded, st);
    }
    flush_stats(a.out.count + (size_t)(wave_global_id() % kQueueClasses) * kCounterStride, n_shaded);
}

// Tree path, bounce 0: the camera rays' packet walk (k_trace_primary_packet) and the shading of the vertices it finds in one
// kernel, like the small-scene path's k_trace_shade<FIRST>: the hit records (32 B per path written and read back) stay in
// registers, and the shade stage's streaming writes -- three planes and two queue entries per path, the HBM-bound part --
// overlap the packet walk's arithmetic of the other waves.  Only the AOV slot's hits are stored (launch_geo_aov reads them).
#ifndef CAP_PS_BLOCKS
#define CAP_PS_BLOCKS 8  // workgroups per CU (residency sweep 4 ... 8: 3.7, 3.25, 3.05, 2.86, 2.81 ms: the packet walk wants waves more than registers)
#endif
template <bool EXT>
__global__ __launch_bounds__(kBlock, CAP_PS_BLOCKS) void k_primary_shade(BvhDev bvh, ShadeArgs a, float4* hits_out)
{
    __shared__ uint32_t   lds_wstack[(kBlock / 64) * kPacketStack];
    __shared__ FrameConst lds_frames[kMaxFrameSlots];
    stage_frames(a, lds_frames);  // ends with the workgroup barrier
    uint32_t*      wstack   = lds_wstack + (threadIdx.x >> 6) * kPacketStack;
    const uint32_t Ppad     = a.screen.pixels_padded;
    const uint32_t cps      = Ppad >> 6;
    const uint32_t chunks   = cps * a.n_slots;
    const uint32_t my_class = wave_global_id() % kQueueClasses;
    uint32_t       n_shaded = 0;
    uint32_t       grab     = grab_issue(a.work, my_class);
    Stamps         st;
    st.start();
    while (true)
    {
        const uint32_t chunk = class_chunk(grab_value(grab), my_class);  // chunk_class(chunk) == my_class: the paths' class, as k_shade<FIRST> assigns it
        if (chunk >= chunks) break;
        grab = grab_issue(a.work, my_class);
        const uint32_t slot = chunk / cps;  // wave-uniform
        const uint32_t pl   = (chunk - slot * cps) * 64 + (threadIdx.x & 63u);
        uint32_t       x = 0, y = 0;
        const bool     alive = local_pixel_to_xy(a.screen, pl, x, y);
        const Ray      r     = make_ray(mk3(a.cam.position[0], a.cam.position[1], a.cam.position[2]),
                                        alive ? primary_dir(a.cam, a.screen, lds_frames[slot], x, y) : mk3(0.f, 0.f, 1.f), 0.0f, kPrimaryFar);
        float          t, u, v;
        uint32_t       gid;
        traverse_closest_packet(bvh, r, alive, wstack, t, u, v, gid);
        const float4 hit = make_float4(alive ? u : 0.0f, alive ? v : 0.0f, u2f(gid), alive ? t : kPrimaryFar);
        if (slot == a.aov_slot) hits_out[(size_t)slot * Ppad + pl] = hit;
        const uint32_t pid = (slot << kPidShift) | pl;
        const ShadePre pre = shade_prefetch<EXT>(a, lds_frames, true, pid);
        if constexpr (EXT)
            shade_vertex_ext<true>(a, a.scene.shade_tris, pre, my_class, pid, hit, mk3(1.0f, 1.0f, 1.0f), r.d, n_shaded);
        else
            shade_vertex<true, false, false, true>(a, a.scene.shade_tris, pre, my_class, pid, hit, mk3(1.0f, 1.0f, 1.0f), n_shaded, st);
    }
    flush_stats(a.out.count + (size_t)my_class * kCounterStride, n_shaded);
}

bool launch_primary_shade(const LaunchCfg& cfg, const BvhDev& bvh, const ShadeArgs& args, float4* hits, bool ext)
{
    const bool off = cfg.sw_on(SW_NO_PRIMARY_FUSE) || cfg.sw_on(SW_NO_PACKET);  // A/B switches
    if (off || cfg.stack_entries == 0 || !args.work || bvh.tri_count < 2 || !cfg.cu_count) return false;
    const uint32_t chunks = (args.screen.pixels_padded >> 6) * args.n_slots;
    uint32_t       gx     = (chunks + 3) / 4;
    const uint32_t cap    = ext ? resident_grid<k_primary_shade<true>>(cfg, ~0u) : resident_grid<k_primary_shade<false>>(cfg, ~0u);
    if (gx > cap) gx = cap;
    if (gx == 0) gx = 1;
    if (ext)
        hipLaunchKernelGGL(k_primary_shade<true>, dim3(gx), dim3(kBlock), 0, cfg.stream, bvh, args, hits);
    else
        hipLaunchKernelGGL(k_primary_shade<false>, dim3(gx), dim3(kBlock), 0, cfg.stream, bvh, args, hits);
    return true;
}

void launch_shade(const LaunchCfg& cfg, const ShadeArgs& args, bool ext, bool feedback)
{
    if (args.bounce == 0)
    {
        const uint32_t chunks = args.screen.pixels_padded >> 6;
        uint32_t       gx     = (chunks + 3) / 4;
        if (gx > cfg.grid_blocks) gx = cfg.grid_blocks;
        if (gx == 0) gx = 1;
        if (ext)
            hipLaunchKernelGGL((k_shade<true, true>), dim3(gx, args.n_slots), dim3(kBlock), 0, cfg.stream, args);
        else
            hipLaunchKernelGGL((k_shade<true, false>), dim3(gx, args.n_slots), dim3(kBlock), 0, cfg.stream, args);
    }
    else
    {
        // static chunk assignment: the grid must be resident at once (resident_grid())
        const uint32_t want = queue_grid(cfg, args.max_count);
        if (ext)
            hipLaunchKernelGGL((k_shade<false, true>), dim3(resident_grid<k_shade<false, true, false>>(cfg, want)), dim3(kBlock), 0, cfg.stream, args);
        else if (feedback)
            hipLaunchKernelGGL((k_shade<false, false, true>), dim3(resident_grid<k_shade<false, false, true>>(cfg, want)), dim3(kBlock), 0,
                               cfg.stream, args);
        else
            hipLaunchKernelGGL((k_shade<false, false>), dim3(resident_grid<k_shade<false, false, false>>(cfg, want)), dim3(kBlock), 0, cfg.stream,
                               args);
    }
}

// Fused stage of the small-scene path: closest-hit (exhaustive, wave-uniform) + shading of the vertex it finds, in one pass over
// the ray queue.  The hit record never travels through HBM and the shading stage's memory latency hides under the ALU-bound
// triangle loop of the other waves.  FIRST generates the camera ray instead of reading a queue entry (rt_primary_visibility).
#ifndef CAP_TS_FIRST
#define CAP_TS_FIRST 5  // workgroups per CU the bounce-0 kernel is register-allocated for
#endif
#ifndef CAP_TS_EXT
#define CAP_TS_EXT 6  // ... and the EXT model's bounce >= 1 kernel
#endif
#ifndef CAP_TS_NEXT
#define CAP_TS_NEXT 6  // ... and the bounce >= 1 kernel (8 fits in 64 VGPRs without spills but measured 8 % slower)
#endif
// LDS: scenes of at most kExhaustiveMax triangles keep their shading records (96 B each) and intersection records in LDS
// (<= 10 KB per workgroup), so the gathers by hit triangle after the loop are ds_reads instead of a global round trip.
template <bool FIRST, bool EXT, bool FB = false, bool LDS = false>
__global__ __launch_bounds__(kBlock, FB ? 4 : (EXT ? (FIRST ? 5 : CAP_TS_EXT) : (FIRST ? CAP_TS_FIRST : CAP_TS_NEXT))) void k_trace_shade(BvhDev bvh, ShadeArgs a)
{
    constexpr bool CARRY    = !EXT;
    const uint32_t Ppad     = a.screen.pixels_padded;
    // FIRST: the identity queue of the whole batch, chunk = slot * (Ppad / 64) + 64-pixel group.  Otherwise: chunk slots of the
    // input queue.  Either way the grid is persistent (the LDS tables are staged once per workgroup, not once per frame slot).
    const uint32_t cps      = Ppad >> 6;
    const uint32_t chunks   = FIRST ? cps * a.n_slots : (a.in.class_capacity >> 6) * kQueueClasses;
    uint32_t       n_shaded = 0;
    __shared__ FrameConst lds_frames[kMaxFrameSlots];
    __shared__ float4     lds_shade[LDS ? kShadeRec * kExhaustiveMax : 1];
    __shared__ float4     lds_rec[LDS ? 4 * kExhaustiveMax : 1];
    // the producer-side shadow probe (ShadeArgs::inline_probe): the probe pair's PairPre rows per frame slot
    constexpr bool        PROBE = !EXT && !FB && LDS;
    __shared__ float4     lds_probe[PROBE ? 2 * kMaxFrameSlots : 1];
    __shared__ float      lds_pscore[PROBE ? kExhaustiveMax / 2 : 1];
    __shared__ uint32_t   lds_probe_k;
    __shared__ float4     lds_ring[(PROBE && !FIRST) ? (kBlock / 64) * kWaveRing : 1];  // per wave: (origin, path id) of its parked shadow rays
    uint32_t              n_probed = 0;
    constexpr bool        ORG = FIRST && LDS;  // camera rays of a small scene: per-pair origin terms from a table (pair_scaled<ORG>)
    __shared__ float4     lds_org[ORG ? kExhaustiveMax : 1];
    // ORG: pixel bounds (x0, y0, x1, y1) of every fan pair as the launch's camera sees it, grown by two pixels (the sub-pixel
    // jitter of the frames and the rounding of the projection); a pair with a vertex at or behind the camera plane covers the
    // screen.  A tile of camera rays only tests the pairs whose bounds overlap it: a ray can only hit a quad through a sample
    // point inside the quad's projection, so the pairs left out are missed by all 64 rays -- same hits, same bits.
    __shared__ float4     lds_bounds[ORG ? kExhaustiveMax / 2 : 1];
    // EXT model: materials, light table and per-light records (ExtTables) -- staged when they fit
    constexpr bool        XT = EXT && LDS;
    __shared__ float      lds_mat[XT ? 12 * kExhaustiveMax : 1];
    __shared__ float      lds_lcdf[XT ? kExtLightsMax : 1];
    __shared__ float4     lds_lrec[XT ? 4 * kExtLightsMax : 1];
    const bool            ext_tabs = XT && a.scene.material_count <= kExhaustiveMax && a.scene.light_count <= kExtLightsMax;  // wave-uniform
    if (LDS)
    {
        const uint32_t n = bvh.tri_count <= kExhaustiveMax ? bvh.tri_count : kExhaustiveMax;
        for (uint32_t k = threadIdx.x; k < kShadeRec * n; k += kBlock) lds_shade[k] = a.scene.shade_tris[k];
        for (uint32_t k = threadIdx.x; k < 4 * n; k += kBlock) lds_rec[k] = bvh.tris_by_id[k];
        if (XT && ext_tabs)
        {
            const float* ms = reinterpret_cast<const float*>(a.scene.materials);
            for (uint32_t k = threadIdx.x; k < 12u * a.scene.material_count; k += kBlock) lds_mat[k] = ms[k];
            for (uint32_t k = threadIdx.x; k < a.scene.light_count; k += kBlock)
            {
                lds_lcdf[k] = a.scene.light_cdf[k];
                const float4* lt = a.scene.shade_tris + kShadeRec * (size_t)a.scene.light_tris[k];
                const float4  l0 = lt[0], l1 = lt[1], l2 = lt[2];
                const v3      q0 = mk3(l0.x, l0.y, l0.z), q1 = mk3(l1.x, l1.y, l1.z), q2 = mk3(l2.x, l2.y, l2.z);
                const v3      nl = normalize3(cross3(q1 - q0, q2 - q0));  // as shade_vertex_ext computes it per vertex without the table
                const MaterialDev lm = a.scene.materials[f2u(lt[6].x)];
                lds_lrec[4 * k]     = make_float4(q0.x, q0.y, q0.z, lm.ke[0]);
                lds_lrec[4 * k + 1] = make_float4(q1.x, q1.y, q1.z, lm.ke[1]);
                lds_lrec[4 * k + 2] = make_float4(q2.x, q2.y, q2.z, lm.ke[2]);
                lds_lrec[4 * k + 3] = make_float4(nl.x, nl.y, nl.z, 0.f);
            }
        }
        if (ORG)
        {
            const v3     o  = mk3(a.cam.position[0], a.cam.position[1], a.cam.position[2]);
            const float* fp = reinterpret_cast<const float*>(bvh.fan_pairs);
            for (uint32_t k = threadIdx.x; k < bvh.fan_pair_count && 2 * k + 1 < kExhaustiveMax; k += kBlock)
            {
                const float* rec  = fp + 20 * (size_t)k;  // (v0, e1, e2, e3, nA, nB, id, 0)
                const v3     tvec = o - mk3(rec[0], rec[1], rec[2]);
                lds_org[2 * k]     = make_float4(tvec.x, tvec.y, tvec.z, dot3(tvec, mk3(rec[12], rec[13], rec[14])));
                lds_org[2 * k + 1] = make_float4(dot3(tvec, mk3(rec[15], rec[16], rec[17])), 0.f, 0.f, 0.f);
                // screen bounds: pixel = ((f * (d.right) / (d.forward)) / sensor + 0.5) * extent  (inverse of primary_dir, camera.h:39-63)
                const v3 R = mk3(a.cam.right[0], a.cam.right[1], a.cam.right[2]), U = mk3(a.cam.up[0], a.cam.up[1], a.cam.up[2]),
                         F = mk3(a.cam.forward[0], a.cam.forward[1], a.cam.forward[2]);
                float x0 = 3.0e38f, y0 = 3.0e38f, x1 = -3.0e38f, y1 = -3.0e38f;
                bool  behind = false;
                for (int e = 0; e < 4; ++e)
                {
                    const v3    d = e == 0 ? tvec * -1.0f : (mk3(rec[3 * e], rec[3 * e + 1], rec[3 * e + 2]) - tvec);  // vertex - camera
                    const float z = dot3(d, F);
                    behind |= !(z > 1e-4f);
                    const float px = ((a.cam.focal_length * dot3(d, R) / z) / a.cam.sensor_x + 0.5f) * (float)a.screen.width;
                    const float py = ((a.cam.focal_length * dot3(d, U) / z) / a.cam.sensor_y + 0.5f) * (float)a.screen.height;
                    x0 = fminf(x0, px), x1 = fmaxf(x1, px), y0 = fminf(y0, py), y1 = fmaxf(y1, py);
                }
                const bool usable = a.cull_camera_pairs != 0u && !behind && x0 == x0 && y0 == y0 && x1 == x1 && y1 == y1;
                lds_bounds[k] = usable ? make_float4(x0 - 2.0f, y0 - 2.0f, x1 + 2.0f, y1 + 2.0f) : make_float4(-3.0e38f, -3.0e38f, 3.0e38f, 3.0e38f);
            }
        }
    }
    if (PROBE && a.inline_probe)
    {
        // the pair whose four vertices reach farthest along the batch's first light direction (k_trace_any_small's first probe)
        const uint32_t np = bvh.fan_pair_count;  // 1 .. kExhaustiveMax / 2, checked by the host
        const float*   fp = reinterpret_cast<const float*>(bvh.fan_pairs);
        if (threadIdx.x < np)
        {
            const float* rec = fp + 20 * (size_t)threadIdx.x;
            const v3     L   = mk3(a.frames[0].light_dir[0], a.frames[0].light_dir[1], a.frames[0].light_dir[2]);
            const v3     v0  = mk3(rec[0], rec[1], rec[2]);
            float        sc  = dot3(v0, L);
            for (int e = 0; e < 3; ++e) sc += dot3(v0 + mk3(rec[3 + 3 * e], rec[4 + 3 * e], rec[5 + 3 * e]), L);
            lds_pscore[threadIdx.x] = sc;
        }
        __syncthreads();
        if (threadIdx.x < np)
        {
            const float sc   = lds_pscore[threadIdx.x];
            uint32_t    rank = 0;
            for (uint32_t j = 0; j < np; ++j)
            {
                const float o = lds_pscore[j];
                rank += (o > sc || (o == sc && j < threadIdx.x)) ? 1u : 0u;
            }
            if (rank == 0) lds_probe_k = threadIdx.x;
        }
        __syncthreads();
        const float* rec = fp + 20 * (size_t)lds_probe_k;
        for (uint32_t sl = threadIdx.x; sl < a.n_slots && sl < kMaxFrameSlots; sl += kBlock)
        {
            const v3 d = mk3(a.frames[sl].light_dir[0], a.frames[sl].light_dir[1], a.frames[sl].light_dir[2]);
            lds_probe[2 * sl]     = tri_pre(d, mk3(rec[12], rec[13], rec[14]), kRayEps, kRayFar);
            lds_probe[2 * sl + 1] = tri_pre(d, mk3(rec[15], rec[16], rec[17]), kRayEps, kRayFar);
        }
    }
    stage_frames(a, lds_frames);  // ends with the workgroup barrier
    ProbeArgs probe;
    if (PROBE && a.inline_probe) probe.rows = lds_probe, probe.pairs = bvh.fan_pairs, probe.k = lds_probe_k;
    // The probe's survivors stay with the wave that found them (ShadeArgs::wave_ring): parked in its own 128-entry ring and traced
    // 64 at a time between chunks -- the any-hit kernel's loop on full waves, without its launch, its queue round trip, or any
    // other wave.  No plane entry is shared: within one launch a path either escapes (sky term) or has a vertex (this shadow
    // ray), and the previous bounce's additions were made by the previous launch.
    uint32_t ring_head = 0, ring_n = 0;  // wave-uniform
    if (PROBE && !FIRST && a.inline_probe && a.wave_ring)  // (bounce 0: more survivors per chunk, and a kernel short of registers: 2.4 -> 2.9 ms for the 0.35 ms of its any-hit launch)
    {
        probe.ring_org = lds_ring + (threadIdx.x >> 6) * kWaveRing;  // origins + path ids in LDS, the contributions in this wave's
        probe.ring_con = a.shadow.contrib_pid + (size_t)wave_global_id() * kWaveRing;  // slice of the shadow queue's memory
    }
    auto trace_ring = [&](uint32_t count) {
        // The ring is a cross-lane hand-off inside one wave: lane i stored entry `pos` (LDS origin, global contribution), lane j
        // loads it here.  Commit 6a9000f put a scheduling barrier here after reading the ISA, not after a failure: nothing but
        // may-alias analysis kept the compiler from hoisting these loads above the stores of the inlined shade_vertex.  The memory
        // model's statement of the same thing: release after the stores (shade_vertex), acquire before the loads.
        wave_handoff();
        const uint32_t lane_ = threadIdx.x & 63u;
        const bool     on    = lane_ < count;
        const uint32_t pos   = (ring_head + lane_) & (kWaveRing - 1u);
        float4         o     = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) o = probe.ring_org[pos];
        const uint32_t spid = f2u(o.w);
        const bool     good = on && (spid >> kPidShift) < a.n_slots && (spid & kPidMask) < Ppad;
        const FrameConst& fc = lds_frames[good ? (spid >> kPidShift) : 0u];
        const Ray  sr = make_ray(mk3(o.x, o.y, o.z), mk3(fc.light_dir[0], fc.light_dir[1], fc.light_dir[2]), kRayEps, good ? kRayFar : kRayEps);
        // what the probe left over is mostly unoccluded, and the path id came out of LDS: the contribution and the plane entry are
        // requested before the test and arrive under it
        float4* const target = a.bounce == 0 ? a.planes.direct : a.planes.color;
        const size_t  idx    = good ? (size_t)(spid >> kPidShift) * Ppad + (spid & kPidMask) : 0;
        float4        c = make_float4(0.f, 0.f, 0.f, 0.f), cur = c;
        if (good) c = probe.ring_con[pos], cur = target[idx];
        const bool occluded = exhaustive_any<false>(bvh, sr);
        // lighting.h:57-60: unoccluded -> the contribution evaluated at shading time is added
        if (good && !occluded) target[idx] = make_float4(cur.x + c.x, cur.y + c.y, cur.z + c.z, cur.w);
        ring_head = (ring_head + count) & (kWaveRing - 1u);
        ring_n -= count;
    };
    const float4* shade_tab = LDS ? lds_shade : a.scene.shade_tris;
    const float4* rec_tab   = LDS ? lds_rec : bvh.tris_by_id;
    ExtTables     xtabs;
    if (XT && ext_tabs) xtabs.materials = reinterpret_cast<const MaterialDev*>(lds_mat), xtabs.light_cdf = lds_lcdf, xtabs.light_rec = lds_lrec;
    Stamps st;
    st.start();
    // Chunk slots from the class's work counter, like the any-hit kernel (with the priorities below: bounce 0 4.6 -> 4.1 ms,
    // bounce >= 1 unchanged; before them it cost the bounce >= 1 kernel 7 %).  As there, the class's length is read once and
    // the next grab is issued after the entry loads (see k_trace_any).
    const uint32_t my_class = wave_global_id() % kQueueClasses;
    const uint32_t lane     = threadIdx.x & 63u;
    uint32_t       n_class  = 0;
    if (!FIRST)
    {
        n_class = a.in.count[my_class * kCounterStride];
        n_class = n_class < a.in.class_capacity ? n_class : a.in.class_capacity;
    }
    uint32_t grab = grab_issue(a.work, my_class);
    while (true)
    {
        const uint32_t j = grab_value(grab);  // slot j of this class
        uint32_t       i, pid = 0, slot = 0;
        const uint32_t klass = my_class;      // the path's class for its whole life
        bool           active;
        v3             thr = mk3(1.0f, 1.0f, 1.0f);
        Ray            r   = make_ray(mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 1.f), 0.0f, 0.0f);  // empty interval: hits nothing
        float          carried_r1 = 0.f, carried_r2 = 0.f;
        if (FIRST)
        {
            const uint32_t chunk = class_chunk(j, my_class);
            if (chunk >= chunks) break;  // (only in or past the last block of 64 chunks: every earlier block holds each class once)
            grab   = grab_issue(a.work, my_class);
            slot   = chunk / cps;  // wave-uniform
            i      = (chunk - slot * cps) * 64 + lane;
            active = true;
            pid    = (slot << kPidShift) | i;
            uint32_t x, y;
            if (local_pixel_to_xy(a.screen, i, x, y))
                r = make_ray(mk3(a.cam.position[0], a.cam.position[1], a.cam.position[2]),
                             primary_dir(a.cam, a.screen, lds_frames[slot], x, y), 0.0f, kPrimaryFar);
        }
        else
        {
            if (j * 64u >= n_class) break;  // past the end of this class's sub-queue
            active = j * 64u + lane < n_class;
            i      = my_class * a.in.class_capacity + j * 64u + lane;
            // extension rays: tmin / tmax are constants (rt_indirect.hlsl:154-157); with CARRY the .w slots hold the sample
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f), d = make_float4(0.f, 0.f, 1.f, 0.f), tp = make_float4(1.f, 1.f, 1.f, 0.f);
            if (active) o = a.in.org_tmin[i], d = a.in.dir_tmax[i], tp = a.in.thr_pid[i];
            grab = grab_issue(a.work, my_class);
            if (active)
            {
                r   = make_ray(mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), kRayEps, kRayFar);
                thr = mk3(tp.x, tp.y, tp.z), pid = f2u(tp.w);
                carried_r1 = o.w, carried_r2 = d.w;
            }
        }
        float    t, u, v;
        uint32_t gid;
        STAMP(st, 0, true);  // queue entry arrived
        // The triangle loop is the long, purely arithmetic phase; everything around it (queue reads, shading with its LDS gathers,
        // the append atomic, the stores) is short and latency-bound.  Raising the wave's priority outside the loop lets those
        // phases issue ahead of other waves' loops, so more memory operations are in flight per SIMD (closest 18.1 -> 17.5 ms;
        // the opposite assignment: no gain).
        uint32_t pair_mask = ~0u;
        if (ORG)
        {
            // the tile of this chunk against the pairs' screen bounds: lane k answers for pair k
            uint32_t tx = 0, ty = 0;
            (void)local_pixel_to_xy(a.screen, i & ~63u, tx, ty);  // first pixel of the 8x8 tile
            const float4 b    = lds_bounds[lane < kExhaustiveMax / 2 ? lane : 0u];
            const bool   over = lane < bvh.fan_pair_count && b.x < (float)(tx + 8u) && b.z >= (float)tx && b.y < (float)(ty + 8u) && b.w >= (float)ty;
            pair_mask         = (uint32_t)__ballot(over);
        }
        __builtin_amdgcn_s_setprio(0);
        exhaustive_closest<ORG, !LDS>(bvh, rec_tab, r, t, u, v, gid, lds_org, pair_mask);
        __builtin_amdgcn_s_setprio(3);
        STAMP(st, 1, true);  // triangle loop + winner's record
        const ShadePre pre = shade_prefetch<EXT, FIRST, CARRY>(a, lds_frames, active, pid, carried_r1, carried_r2);
        if (FIRST && slot == a.aov_slot)
        {
            // rt_primary_visibility.hlsl:46: (uv, asfloat(InstanceID), asfloat(PrimitiveIndex)); a miss keeps uv = 0, ids = ~0u
            float4 g = make_float4(0.f, 0.f, u2f(kInvalidId), u2f(kInvalidId));
            if (gid != kInvalidId)
            {
                const uint4 id = a.scene.tri_ids[gid];
                g              = make_float4(u, v, u2f(id.x), u2f(id.y));
            }
            a.planes.aov_geo[i] = g;
        }
        if constexpr (EXT)
        {
            if (a.inline_nee)  // wave-uniform
            {
                v3 acc = mk3(0.f, 0.f, 0.f);
                if (!FIRST && active)
                {
                    const float4 q = a.in.acc[i];
                    acc            = mk3(q.x, q.y, q.z);
                }
                shade_vertex_ext<FIRST, true>(a, shade_tab, pre, klass, pid, make_float4(u, v, u2f(gid), t), thr, r.d, n_shaded, &bvh, acc, xtabs);
            }
            else
                shade_vertex_ext<FIRST>(a, shade_tab, pre, klass, pid, make_float4(u, v, u2f(gid), t), thr, r.d, n_shaded, nullptr, mk3(0.f, 0.f, 0.f), xtabs);
        }
        else
        {
            shade_vertex<FIRST, FB, CARRY, false, PROBE>(a, shade_tab, pre, klass, pid, make_float4(u, v, u2f(gid), t), thr, n_shaded, st, probe, &n_probed,
                                                         ring_head, &ring_n);
            if (PROBE && ring_n >= 64u) trace_ring(64u);
        }
        STAMP(st, 4, false);  // stores issued
    }
    if (!FIRST && !EXT && !FB) st.flush();
#ifdef CAP_STAMPS
    if (!FIRST && !EXT && !FB && (threadIdx.x & 63u) == 0 && wave_global_id() < 16384)
    {
        g_wave_times[2 * wave_global_id() + 0] = st.t_begin;
        g_wave_times[2 * wave_global_id() + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (PROBE && ring_n != 0u) trace_ring(ring_n);  // what is left in this wave's ring
    flush_stats(a.out.count + (size_t)my_class * kCounterStride, n_shaded, n_probed);
}

#ifdef CAP_STAMPS
extern "C" int cap_debug_stamps(unsigned long long* out, int reset)
{
    if (reset == 2) return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), 2 * 16384 * sizeof(unsigned long long));
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset)
    {
        unsigned long long z[16] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
    }
    return (int)e;
}
#endif

void launch_trace_shade(const LaunchCfg& cfg, const BvhDev& bvh, const ShadeArgs& args, bool ext, bool feedback)
{
    const bool lds = bvh.tri_count <= kExhaustiveMax;
    if (args.bounce == 0)
    {
        const uint32_t chunks = (args.screen.pixels_padded >> 6) * args.n_slots;
        uint32_t       gx     = (chunks + 3) / 4;
        if (gx > cfg.grid_blocks) gx = cfg.grid_blocks;
        if (gx == 0) gx = 1;
        const dim3 grid(gx), block(kBlock);
        if (ext && lds)
            hipLaunchKernelGGL((k_trace_shade<true, true, false, true>), grid, block, 0, cfg.stream, bvh, args);
        else if (ext)
            hipLaunchKernelGGL((k_trace_shade<true, true, false, false>), grid, block, 0, cfg.stream, bvh, args);
        else if (lds)
            hipLaunchKernelGGL((k_trace_shade<true, false, false, true>), grid, block, 0, cfg.stream, bvh, args);
        else
            hipLaunchKernelGGL((k_trace_shade<true, false, false, false>), grid, block, 0, cfg.stream, bvh, args);
        return;
    }
    const dim3 grid(queue_grid(cfg, args.max_count)), block(kBlock);
    if (ext && lds)
        hipLaunchKernelGGL((k_trace_shade<false, true, false, true>), grid, block, 0, cfg.stream, bvh, args);
    else if (ext)
        hipLaunchKernelGGL((k_trace_shade<false, true, false, false>), grid, block, 0, cfg.stream, bvh, args);
    else if (feedback && lds)
        hipLaunchKernelGGL((k_trace_shade<false, false, true, true>), grid, block, 0, cfg.stream, bvh, args);
    else if (feedback)
        hipLaunchKernelGGL((k_trace_shade<false, false, true, false>), grid, block, 0, cfg.stream, bvh, args);
    else if (lds)
        hipLaunchKernelGGL((k_trace_shade<false, false, false, true>), grid, block, 0, cfg.stream, bvh, args);
    else
        hipLaunchKernelGGL((k_trace_shade<false, false, false, false>), grid, block, 0, cfg.stream, bvh, args);
}

// ------------------------------------------------------------------------------------------------
// Accumulate / exchange
// ------------------------------------------------------------------------------------------------
// combine_illumination.hlsl:29 per frame, then a plain running fp32 sum in frame order.
// ALBEDO_IN_W (ShadeArgs::albedo_in_w): no albedo plane; direct.w says which of the four constant albedos the path's first vertex has
template <bool ALBEDO_IN_W>
__global__ __launch_bounds__(kBlock) void k_resolve(Planes planes, uint32_t n_slots, uint32_t Ppad, float4* accum, float kd_untextured)
{
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < Ppad; pl += gridDim.x * kBlock)
    {
        float4 acc = accum[pl];
        for (uint32_t s = 0; s < n_slots; ++s)
        {
            const size_t idx = (size_t)s * Ppad + pl;
            const float4 c = planes.color[idx], d = planes.direct[idx];
            float4       al;
            if (ALBEDO_IN_W)
            {
                const float k = d.w == 1.0f ? 1.0f : (d.w == 2.0f ? kd_untextured : 0.0f);
                al            = make_float4(k, k, k, 0.f);
            }
            else
                al = planes.albedo[idx];
            acc.x = acc.x + (c.x * al.x + d.x);
            acc.y = acc.y + (c.y * al.y + d.y);
            acc.z = acc.z + (c.z * al.z + d.z);
            acc.w = acc.w + 1.0f;
        }
        accum[pl] = acc;
    }
}

void launch_resolve(const LaunchCfg& cfg, const Planes& planes, uint32_t n_slots, uint32_t Ppad, float4* accum, bool albedo_in_w,
                    float kd_untextured)
{
    uint32_t g = (Ppad + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (albedo_in_w)
        hipLaunchKernelGGL(k_resolve<true>, dim3(g ? g : 1), dim3(kBlock), 0, cfg.stream, planes, n_slots, Ppad, accum, kd_untextured);
    else
        hipLaunchKernelGGL(k_resolve<false>, dim3(g ? g : 1), dim3(kBlock), 0, cfg.stream, planes, n_slots, Ppad, accum, kd_untextured);
}

__global__ __launch_bounds__(kBlock) void k_untile(ScreenDev sc, const float4* src, const float4* albedo, const float4* direct,
                                                   int kind, float4* image)
{
    const uint32_t n = sc.local_tiles * kTilePixels;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < n; pl += gridDim.x * kBlock)
    {
        uint32_t x, y;
        if (!local_pixel_to_xy(sc, pl, x, y)) continue;
        float4 v = src[pl];
        if (kind == 1)
        {
            const float4 al = albedo[pl], d = direct[pl];
            // combine_illumination.hlsl:24,29 (indirect.w is forced to 1 before the multiply-add)
            v = make_float4(v.x * al.x + d.x, v.y * al.y + d.y, v.z * al.z + d.z, 1.0f * al.w + d.w);
        }
        else if (kind == 2)
        {
            v = make_float4(v.x / v.w, v.y / v.w, v.z / v.w, v.w);
        }
        image[(size_t)y * sc.width + x] = v;
    }
}

// four tile-ordered planes -> four row-major images in one pass (the reconstruction chain's inputs)
__global__ __launch_bounds__(kBlock) void k_untile4(ScreenDev sc, const float4* s0, const float4* s1, const float4* s2, const float4* s3,
                                                    float4* d0, float4* d1, float4* d2, float4* d3)
{
    const uint32_t n = sc.local_tiles * kTilePixels;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < n; pl += gridDim.x * kBlock)
    {
        uint32_t x, y;
        if (!local_pixel_to_xy(sc, pl, x, y)) continue;
        const size_t o = (size_t)y * sc.width + x;
        if (s0) d0[o] = s0[pl];
        d1[o] = s1[pl], d2[o] = s2[pl], d3[o] = s3[pl];
    }
}

void launch_untile4(const LaunchCfg& cfg, const ScreenDev& screen, const float4* s0, const float4* s1, const float4* s2, const float4* s3,
                    float4* d0, float4* d1, float4* d2, float4* d3)
{
    uint32_t g = (screen.local_tiles * kTilePixels + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_untile4, dim3(g ? g : 1), dim3(kBlock), 0, cfg.stream, screen, s0, s1, s2, s3, d0, d1, d2, d3);
}

void launch_untile(const LaunchCfg& cfg, const ScreenDev& screen, const float4* src, const float4* albedo, const float4* direct,
                   int plane_kind, float4* image)
{
    uint32_t g = (screen.local_tiles * kTilePixels + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_untile, dim3(g ? g : 1), dim3(kBlock), 0, cfg.stream, screen, src, albedo, direct, plane_kind, image);
}

// row-major image -> this shard's tile-ordered buffer (the inverse of k_untile, kind 0); padding lanes and other shards' pixels: 0
__global__ __launch_bounds__(kBlock) void k_tile(ScreenDev sc, const float4* image, float4* dst)
{
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < sc.pixels_padded; pl += gridDim.x * kBlock)
    {
        uint32_t x, y;
        dst[pl] = local_pixel_to_xy(sc, pl, x, y) ? image[(size_t)y * sc.width + x] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

void launch_tile(const LaunchCfg& cfg, const ScreenDev& screen, const float4* image, float4* dst)
{
    uint32_t g = (screen.pixels_padded + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_tile, dim3(g ? g : 1), dim3(kBlock), 0, cfg.stream, screen, image, dst);
}

__global__ __launch_bounds__(kBlock) void k_tiles_mean(const float4* accum, uint32_t Ppad, float4* dst)
{
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < Ppad; pl += gridDim.x * kBlock)
    {
        const float4 v = accum[pl];
        dst[pl] = v.w > 0.0f ? make_float4(v.x / v.w, v.y / v.w, v.z / v.w, v.w) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

void launch_tiles_mean(const LaunchCfg& cfg, const float4* accum, uint32_t Ppad, float4* dst)
{
    uint32_t g = (Ppad + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_tiles_mean, dim3(g ? g : 1), dim3(kBlock), 0, cfg.stream, accum, Ppad, dst);
}

// gathered: [shard][shard_stride >= Ppad] tile-ordered pixels -> row-major image
__global__ __launch_bounds__(kBlock) void k_assemble(ScreenDev sc, const float4* gathered, uint32_t shard_count, size_t shard_stride, float4* image)
{
    const uint32_t total = sc.tile_count * kTilePixels;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock)
    {
        const uint32_t gt = i >> 6, w = i & 63u;
        const uint32_t shard = gt % shard_count, lt = gt / shard_count;
        const uint32_t ty = gt / sc.tiles_x, tx = gt - ty * sc.tiles_x;
        const uint32_t x = tx * kTileDim + (w & 7u), y = ty * kTileDim + (w >> 3);
        if (x < sc.width && y < sc.height)
            image[(size_t)y * sc.width + x] = gathered[(size_t)shard * shard_stride + lt * kTilePixels + w];
    }
}

void launch_assemble(const LaunchCfg& cfg, const ScreenDev& screen, const float4* gathered, uint32_t shard_count, float4* image,
                     size_t shard_stride)
{
    uint32_t g = (screen.tile_count * kTilePixels + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_assemble, dim3(g ? g : 1), dim3(kBlock), 0, cfg.stream, screen, gathered, shard_count,
                       shard_stride ? shard_stride : (size_t)screen.pixels_padded, image);
}

// rt_primary_visibility.hlsl:46: (uv, asfloat(InstanceID), asfloat(PrimitiveIndex)); a miss keeps uv = 0, ids = ~0u (:41-43)
__global__ __launch_bounds__(kBlock) void k_geo_aov(SceneDev scene, const float4* hits, uint32_t Ppad, float4* out)
{
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < Ppad; pl += gridDim.x * kBlock)
    {
        const float4   h   = hits[pl];
        const uint32_t gid = f2u(h.z);
        if (gid == kInvalidId)
            out[pl] = make_float4(0.f, 0.f, u2f(kInvalidId), u2f(kInvalidId));
        else
        {
            const uint4 id = scene.tri_ids[gid];
            out[pl]        = make_float4(h.x, h.y, u2f(id.x), u2f(id.y));
        }
    }
}

void launch_geo_aov(const LaunchCfg& cfg, const SceneDev& scene, const float4* hits_slot, uint32_t Ppad, float4* aov_geo)
{
    uint32_t g = (Ppad + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_geo_aov, dim3(g ? g : 1), dim3(kBlock), 0, cfg.stream, scene, hits_slot, Ppad, aov_geo);
}
}  // namespace cap
