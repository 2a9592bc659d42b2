// cap_wide_trace.h — per-lane traversal steps on the compressed 8-wide tree (cap_wide.h), shared by trace8.hip (closest hit with
// lane refill) and kernels.hip (any hit inside k_trace_any).  Device code only.
#pragma once

#include "cap_trace.h"
#include "cap_wide.h"

namespace cap
{
// What the box test needs of a ray.  inv = 1 / d with |d| < 1e-20 replaced by +-1e-20 (box test only, see wide_builder.cpp), from
// v_rcp_f32: the boxes' padding covers its 1-ulp error and the hit rule never depends on the box test, so this need not be
// reproducible on the host.  noi = -(o * inv): plane parameter = fma(plane, inv, noi).
struct WideRay
{
    v3       inv, noi;
    uint32_t octinv;  // 7 - (d.x < 0 | (d.y < 0) << 1 | (d.z < 0) << 2)
    bool     neg_x, neg_y, neg_z;
};

__device__ __forceinline__ WideRay make_wide_ray(v3 o, v3 d)
{
    auto rcp_safe = [](float x) {
        const float s = fabsf(x) < 1e-20f ? __builtin_copysignf(1e-20f, x) : x;
        return __builtin_amdgcn_rcpf(s);
    };
    WideRay w;
    w.inv   = mk3(rcp_safe(d.x), rcp_safe(d.y), rcp_safe(d.z));
    w.noi   = mk3(-(o.x * w.inv.x), -(o.y * w.inv.y), -(o.z * w.inv.z));
    w.neg_x = w.inv.x < 0.0f, w.neg_y = w.inv.y < 0.0f, w.neg_z = w.inv.z < 0.0f;
    w.octinv = 7u - ((w.neg_x ? 1u : 0u) | (w.neg_y ? 2u : 0u) | (w.neg_z ? 4u : 0u));
    return w;
}

// bit i of the result = bit (i ^ k) of the 8-bit mask m, k given by its three bits
__device__ __forceinline__ uint32_t xorperm8(uint32_t m, bool k0, bool k1, bool k2)
{
    uint32_t a = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
    m          = k0 ? a : m;
    a          = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
    m          = k1 ? a : m;
    a          = ((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu);
    m          = k2 ? a : m;
    return m;
}

// A lane's position in the tree: the node group still to visit (children of one node: first-child index + which of them the
// ray's box test hit, in visiting priority) and the triangle group still to test (leaf children of the node just visited).
struct WideCursor
{
    uint32_t g_base, g_mask;           // g_mask: bits 24..31 = hit inner children, bit 24 + (slot ^ octinv); bits 0..7 = imask
    uint32_t t_base, t_hits, t_valid;  // t_hits: bit (k * 8 + slot) = triangle k of leaf child `slot` is due
};

__device__ __forceinline__ void wide_cursor_root(WideCursor& c)
{
    c.g_base = 0u, c.g_mask = 1u << 24;  // "child 0 of a group whose imask is empty" = node 0
    c.t_base = 0u, c.t_hits = 0u, c.t_valid = 0u;
}

// Takes the next inner child out of the node group; returns its node index.  `rest` = the group still has children to visit.
__device__ __forceinline__ uint32_t wide_pick_child(WideCursor& c, uint32_t octinv, bool& rest)
{
    const uint32_t bit = 31u - (uint32_t)__clz((int)c.g_mask);  // >= 24
    c.g_mask &= ~(1u << bit);
    const uint32_t slot = (bit - 24u) ^ octinv;
    rest                = (c.g_mask >> 24) != 0u;
    return c.g_base + (uint32_t)__popc(c.g_mask & 0xffu & ((1u << slot) - 1u));
}

// The 80 B of a node (five 16-B loads from global memory or from an LDS copy).
struct WideNode
{
    float4 h0, h1, q2, q3, q4;
};
__device__ __forceinline__ WideNode wide_node_load(const float4* __restrict__ N)
{
    WideNode n;
    n.h0 = N[0], n.h1 = N[1], n.q2 = N[2], n.q3 = N[3], n.q4 = N[4];
#ifdef CAP_W8_EXTRA_LOADS  // experiment: what would a 128-B node cost in the load path?
    for (int k = 0; k < CAP_W8_EXTRA_LOADS; ++k)
    {
        const float4 x = N[5 + k];
        asm volatile("" ::"v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
    }
#endif
    return n;
}

// Box tests of the eight children of a node; replaces the cursor's groups by the node's.
#if !defined(CAP_W8_NODE_V1) && !defined(CAP_W8_COUNT)  // (the step-counting diagnostic build keeps the round-2 form: it exports the children's entry distances)
// Box tests of the eight children of a node; replaces the cursor's groups by the node's.
// Round 6 form, written against the measured issue rates of docs/experiments.md (74) (tools/micro/valu_cost.hip): per SIMD a wave64
// v_fma / v_sub / v_bitop3 costs 2 cycles, a select, a compare, a two-operand min / max, a byte conversion or a shift 4.  So
//   * the near / far plane words are picked with bit selects (v_bitop3_b32) by the sign words of the ray's 1 / d instead of 24 selects
//     by SGPR mask, and the octant permutation's three conditional swaps likewise;
//   * a child's verdict is the sign of three differences -- exit - entry, tfar - entry, exit - tmin: the box interval is empty, starts
//     behind the ray's end, or ends before its start -- OR-ed by one v_bitop3 and shifted into the mask by one v_alignbit, instead of
//     clamping entry and exit with tmin / tfar (two min / max), a compare, a select and an OR;
// 26 -> 20 instructions per child, 16 of the 24 that remain half-rate are the six byte conversions and the two three-operand min / max.
// Same verdicts for every finite operand: max(entry, tmin) <= min(exit, tfar)  <=>  entry <= exit, entry <= tfar, tmin <= exit (tmin <=
// tfar holds for a live ray); a NaN difference (inf - inf) reads as "hit", which only costs a visit.  The hit rule never looks at boxes.
// ORDER: 0 = front to back along the ray's octant (closest hit needs it: best_t prunes what lies behind), 1 = slot order (no permutation),
// 2 = back to front.  An occlusion query's answer does not depend on the visiting order: see kAnyOrder.
// OCT: -1 = the signs of the ray's direction are per-lane data; 0 .. 7 = every ray of the launch has these (bit 0: d.x negative, bit 1: y,
// bit 2: z -- the reference model's shadow rays all point at one directional light): near / far words and the visiting permutation are
// then compile-time choices, 12 selects, three sign words and up to three conditional swaps per node step fewer.
template <int ORDER = 0, int OCT = -1>
__device__ __forceinline__ void wide_node_test(const WideNode& n, const WideRay& r, float tmin, float tfar, WideCursor& c)
{
    const float4   h0 = n.h0, h1 = n.h1, q2 = n.q2, q3 = n.q3, q4 = n.q4;
    const uint32_t syz = f2u(h1.w);  // the y and z steps' upper halves (powers of two: the lower halves are zero)
    const float    ax = h0.w * r.inv.x, ay = u2f(syz & 0xffff0000u) * r.inv.y, az = u2f(syz << 16) * r.inv.z;
    const float    bx = fmaf(h0.x, r.inv.x, r.noi.x), by = fmaf(h0.y, r.inv.y, r.noi.y), bz = fmaf(h0.z, r.inv.z, r.noi.z);
    // all ones where the direction component is negative (1 / d is never +-0: |d| <= 1): such a ray enters through the high plane
    const uint32_t mx = OCT >= 0 ? ((OCT & 1) ? ~0u : 0u) : (uint32_t)((int32_t)f2u(r.inv.x) >> 31),
                   my = OCT >= 0 ? ((OCT & 2) ? ~0u : 0u) : (uint32_t)((int32_t)f2u(r.inv.y) >> 31),
                   mz = OCT >= 0 ? ((OCT & 4) ? ~0u : 0u) : (uint32_t)((int32_t)f2u(r.inv.z) >> 31);
    const uint32_t lx0 = f2u(q2.x), lx1 = f2u(q2.y), ly0 = f2u(q2.z), ly1 = f2u(q2.w), lz0 = f2u(q3.x), lz1 = f2u(q3.y);
    const uint32_t hx0 = f2u(q3.z), hx1 = f2u(q3.w), hy0 = f2u(q4.x), hy1 = f2u(q4.y), hz0 = f2u(q4.z), hz1 = f2u(q4.w);
    /* m ? a : b, bit by bit: truth-table index = m * 4 + a * 2 + b (a constant mask folds to a or b) */
#define CAP_W8_SEL(m, a, b) (OCT >= 0 ? ((m) ? (a) : (b)) : __builtin_amdgcn_bitop3_b32((m), (a), (b), 0xca))
    const uint32_t nx0 = CAP_W8_SEL(mx, hx0, lx0), nx1 = CAP_W8_SEL(mx, hx1, lx1), fx0 = CAP_W8_SEL(mx, lx0, hx0), fx1 = CAP_W8_SEL(mx, lx1, hx1);
    const uint32_t ny0 = CAP_W8_SEL(my, hy0, ly0), ny1 = CAP_W8_SEL(my, hy1, ly1), fy0 = CAP_W8_SEL(my, ly0, hy0), fy1 = CAP_W8_SEL(my, ly1, hy1);
    const uint32_t nz0 = CAP_W8_SEL(mz, hz0, lz0), nz1 = CAP_W8_SEL(mz, hz1, lz1), fz0 = CAP_W8_SEL(mz, lz0, hz0), fz1 = CAP_W8_SEL(mz, lz1, hz1);
    uint32_t       miss = 0u;  // children 7 .. 0 shifted in from the right: bit s = child s's box interval misses [tmin, tfar]
#define CAP_W8_CHILD(nxw, nyw, nzw, fxw, fyw, fzw, sh)                                                                                    \
    {                                                                                                                                    \
        const float tnx = fmaf((float)(((nxw) >> (sh)) & 0xffu), ax, bx), tny = fmaf((float)(((nyw) >> (sh)) & 0xffu), ay, by),         \
                    tnz = fmaf((float)(((nzw) >> (sh)) & 0xffu), az, bz);                                                                \
        const float tfx = fmaf((float)(((fxw) >> (sh)) & 0xffu), ax, bx), tfy = fmaf((float)(((fyw) >> (sh)) & 0xffu), ay, by),         \
                    tfz = fmaf((float)(((fzw) >> (sh)) & 0xffu), az, bz);                                                                \
        const float tn = fmaxf(fmaxf(tnx, tny), tnz), tf = fminf(fminf(tfx, tfy), tfz);                                                  \
        const uint32_t sg = __builtin_amdgcn_bitop3_b32(f2u(tf - tn), f2u(tfar - tn), f2u(tf - tmin), 0xfe);                             \
        miss = __builtin_amdgcn_alignbit(miss, sg, 31);                                                                                  \
    }
    CAP_W8_CHILD(nx1, ny1, nz1, fx1, fy1, fz1, 24)
    CAP_W8_CHILD(nx1, ny1, nz1, fx1, fy1, fz1, 16)
    CAP_W8_CHILD(nx1, ny1, nz1, fx1, fy1, fz1, 8)
    CAP_W8_CHILD(nx1, ny1, nz1, fx1, fy1, fz1, 0)
    CAP_W8_CHILD(nx0, ny0, nz0, fx0, fy0, fz0, 24)
    CAP_W8_CHILD(nx0, ny0, nz0, fx0, fy0, fz0, 16)
    CAP_W8_CHILD(nx0, ny0, nz0, fx0, fy0, fz0, 8)
    CAP_W8_CHILD(nx0, ny0, nz0, fx0, fy0, fz0, 0)
#undef CAP_W8_CHILD
    const uint32_t masks = f2u(h1.z), imask = masks >> 24, tvalid = masks & 0x00ffffffu;
    const uint32_t h8 = ~miss & 0xffu;
    // an unused slot is told by the masks, not by its planes (no plane content fails the conservative test reliably)
    // the hit inner children in visiting priority: bit i <- bit i ^ octinv (octinv's bits are the negated signs: swap where the sign is +)
    uint32_t m = h8 & imask;
    if (ORDER != 1)
    {
        // (back to front: the complementary octant, i.e. the swaps where the sign is NEGATIVE)
        uint32_t a = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
        m          = ORDER == 0 ? CAP_W8_SEL(mx, m, a) : CAP_W8_SEL(mx, a, m);
        a          = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
        m          = ORDER == 0 ? CAP_W8_SEL(my, m, a) : CAP_W8_SEL(my, a, m);
        a          = ((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu);
        m          = ORDER == 0 ? CAP_W8_SEL(mz, m, a) : CAP_W8_SEL(mz, a, m);
    }
#undef CAP_W8_SEL
    c.g_base  = f2u(h1.x);
    c.g_mask  = (m << 24) | imask;
    c.t_base  = f2u(h1.y);
    c.t_valid = tvalid;
    c.t_hits  = __builtin_amdgcn_perm(h8, h8, 0x0c000000u) & tvalid;  // h8 in the three low bytes (one v_perm_b32)
}
#else
#ifdef CAP_W8_COUNT  // diagnostic build: the children's entry distances, by slot
#define CAP_W8_TN_ARG , float* tn_out = nullptr
#define CAP_W8_TN_OUT(slot, tn) if (tn_out) tn_out[slot] = tn;
#else
#define CAP_W8_TN_ARG
#define CAP_W8_TN_OUT(slot, tn)
#endif
template <int ORDER = 0, int OCT = -1>  // (the round-2 form always orders front to back and reads the signs per lane)
__device__ __forceinline__ void wide_node_test(const WideNode& n, const WideRay& r, float tmin, float tfar, WideCursor& c CAP_W8_TN_ARG)
{
    const float4   h0 = n.h0, h1 = n.h1, q2 = n.q2, q3 = n.q3, q4 = n.q4;
    const uint32_t syz = f2u(h1.w);  // the y and z steps' upper halves (powers of two: the lower halves are zero)
    const float    ax = h0.w * r.inv.x, ay = u2f(syz & 0xffff0000u) * r.inv.y, az = u2f(syz << 16) * r.inv.z;
    const float    bx = fmaf(h0.x, r.inv.x, r.noi.x), by = fmaf(h0.y, r.inv.y, r.noi.y), bz = fmaf(h0.z, r.inv.z, r.noi.z);
    // near / far plane words per axis (slots 0..3 and 4..7): a negative direction enters through the high plane
    const uint32_t lx0 = f2u(q2.x), lx1 = f2u(q2.y), ly0 = f2u(q2.z), ly1 = f2u(q2.w), lz0 = f2u(q3.x), lz1 = f2u(q3.y);
    const uint32_t hx0 = f2u(q3.z), hx1 = f2u(q3.w), hy0 = f2u(q4.x), hy1 = f2u(q4.y), hz0 = f2u(q4.z), hz1 = f2u(q4.w);
    const uint32_t nx0 = r.neg_x ? hx0 : lx0, nx1 = r.neg_x ? hx1 : lx1, fx0 = r.neg_x ? lx0 : hx0, fx1 = r.neg_x ? lx1 : hx1;
    const uint32_t ny0 = r.neg_y ? hy0 : ly0, ny1 = r.neg_y ? hy1 : ly1, fy0 = r.neg_y ? ly0 : hy0, fy1 = r.neg_y ? ly1 : hy1;
    const uint32_t nz0 = r.neg_z ? hz0 : lz0, nz1 = r.neg_z ? hz1 : lz1, fz0 = r.neg_z ? lz0 : hz0, fz1 = r.neg_z ? lz1 : hz1;
    uint32_t       h8 = 0u;
#define CAP_W8_CHILD(slot, nxw, nyw, nzw, fxw, fyw, fzw, sh)                                                                             \
    {                                                                                                                                    \
        const float tnx = fmaf((float)(((nxw) >> (sh)) & 0xffu), ax, bx), tny = fmaf((float)(((nyw) >> (sh)) & 0xffu), ay, by),         \
                    tnz = fmaf((float)(((nzw) >> (sh)) & 0xffu), az, bz);                                                                \
        const float tfx = fmaf((float)(((fxw) >> (sh)) & 0xffu), ax, bx), tfy = fmaf((float)(((fyw) >> (sh)) & 0xffu), ay, by),         \
                    tfz = fmaf((float)(((fzw) >> (sh)) & 0xffu), az, bz);                                                                \
        const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin)), tf = fminf(fminf(tfx, tfy), fminf(tfz, tfar));                        \
        h8 |= (tn <= tf) ? (1u << (slot)) : 0u;                                                                                          \
        CAP_W8_TN_OUT(slot, tn)                                                                                                          \
    }
    CAP_W8_CHILD(0, nx0, ny0, nz0, fx0, fy0, fz0, 0)
    CAP_W8_CHILD(1, nx0, ny0, nz0, fx0, fy0, fz0, 8)
    CAP_W8_CHILD(2, nx0, ny0, nz0, fx0, fy0, fz0, 16)
    CAP_W8_CHILD(3, nx0, ny0, nz0, fx0, fy0, fz0, 24)
    CAP_W8_CHILD(4, nx1, ny1, nz1, fx1, fy1, fz1, 0)
    CAP_W8_CHILD(5, nx1, ny1, nz1, fx1, fy1, fz1, 8)
    CAP_W8_CHILD(6, nx1, ny1, nz1, fx1, fy1, fz1, 16)
    CAP_W8_CHILD(7, nx1, ny1, nz1, fx1, fy1, fz1, 24)
#undef CAP_W8_CHILD
    const uint32_t masks = f2u(h1.z), imask = masks >> 24, tvalid = masks & 0x00ffffffu;
    // an unused slot is told by the masks, not by its planes (no plane content fails the conservative test reliably)
    c.g_base  = f2u(h1.x);
    c.g_mask  = (xorperm8(h8 & imask, !r.neg_x, !r.neg_y, !r.neg_z) << 24) | imask;  // octinv's bits are the negated signs
    c.t_base  = f2u(h1.y);
    c.t_valid = tvalid;
    c.t_hits  = __builtin_amdgcn_perm(h8, h8, 0x0c000000u) & tvalid;  // h8 in the three low bytes (one v_perm_b32)
}

#endif  // CAP_W8_NODE_V1 || CAP_W8_COUNT

// Takes the next due triangle out of the triangle group; returns its record index.
__device__ __forceinline__ uint32_t wide_pick_triangle(WideCursor& c)
{
    const uint32_t bit = (uint32_t)__ffs((int)c.t_hits) - 1u;
    c.t_hits &= c.t_hits - 1u;
    return c.t_base + (uint32_t)__popc(c.t_valid & ((1u << bit) - 1u));
}

// Per-lane stack of (g_base, g_mask) pairs: LDS entries first (this lane's column: entry k at lds[k * kBlock]), the rest in the
// thread's slice of BvhDev::stack_spill.  The host enables the wide kernels only when depth - 1 fits both parts together.
template <int LDS_ENTRIES>
struct WideStack
{
    uint2* lds;
    uint2* spill;
    int    sp;
    __device__ __forceinline__ void push(uint32_t base, uint32_t mask)
    {
        if (sp < LDS_ENTRIES)
            lds[sp * kBlock] = make_uint2(base, mask);
        else
            spill[sp - LDS_ENTRIES] = make_uint2(base, mask);
        ++sp;
    }
    __device__ __forceinline__ void pop(WideCursor& c)
    {
        --sp;
        uint2 e;
        if (sp < LDS_ENTRIES)  // explicit branches: a select of the two addresses would become one flat load
            e = lds[sp * kBlock];
        else
            e = spill[sp - LDS_ENTRIES];
        c.g_base = e.x, c.g_mask = e.y;
    }
};
__device__ __forceinline__ uint2* wide_spill_of_thread(const BvhDev& bvh)
{
    return reinterpret_cast<uint2*>(bvh.stack_spill + (size_t)(blockIdx.x * kBlock + threadIdx.x) * kSpillEntries);
}

// Any-hit queries visit the hit children BACK TO FRONT (round 6, docs/experiments.md (83)).  An occlusion query is answered by any occluder,
// so the order is free; front to back -- what the closest-hit kernel needs -- starts with the boxes around the ray's own origin, which a
// ray that leaves a surface enters and leaves without hitting anything, and reaches the structures that do block (whatever stands between
// the scene and the light) last.  Measured, shadow rays of the reference model's light: 262 k hall 4.77 -> 4.03 ms per 32 spp, 4.2 M hall
// 2.11 -> 1.90 per 8 spp, 16.8 M hall 2.63 -> 2.38.  (-DCAP_W8_ANY_ORDER=0 front to back, 1 slot order without the permutation -- for this
// light's octant the same order as 2 and 15 instructions per node step cheaper: 3.88 / 1.84 / 2.34 --, 2 back to front.)
#ifndef CAP_W8_ANY_ORDER
#define CAP_W8_ANY_ORDER 2
#endif
#if !defined(CAP_W8_NODE_V1) && !defined(CAP_W8_COUNT)
constexpr int kAnyOrder = CAP_W8_ANY_ORDER;
#else
constexpr int kAnyOrder = 0;
#endif
__device__ __forceinline__ uint32_t kAnyOct(uint32_t octinv) { return kAnyOrder == 0 ? octinv : (kAnyOrder == 1 ? 0u : octinv ^ 7u); }

// Any hit on the wide tree (lighting.h:48-61 semantics as traverse_any): true when some triangle has tmin < t < tmax.
// lds_words: this lane's column of a [entries][kBlock] uint32 LDS array of STACK_WORDS entries, reused as STACK_WORDS / 2 pairs.
// OCT: see wide_node_test (0 .. 7: the caller has checked that every ray of the launch lies in that octant).
template <int STACK_WORDS, int OCT = -1>
__device__ __forceinline__ bool traverse_any8(const BvhDev& bvh, const Ray& r, uint32_t* lds_words)
{
    // the pair (k) of this lane lives in words 2k and 2k + 1 of its column: two 4-B accesses, conflict-free like the word stack
    struct PairStack
    {
        uint32_t* lds;
        uint2*    spill;
        int       sp;
    } st{lds_words, wide_spill_of_thread(bvh), 0};
    constexpr int kPairs = STACK_WORDS / 2;
    const WideRay w      = make_wide_ray(r.o, r.d);
    WideCursor    c;
    wide_cursor_root(c);
    // every iteration is one load sequence: a lane with a triangle due fetches its 64-B record with the first four of the five
    // loads a node lane needs (see trace8.hip for why load instructions, not lanes, are what costs)
    while (true)
    {
        const bool    tri_lane = c.t_hits != 0u;
        const float4* src;
        if (tri_lane)
            src = bvh.tris8 + 4 * (size_t)wide_pick_triangle(c);
        else
        {
            bool           rest;
            const uint32_t node = wide_pick_child(c, kAnyOct(OCT >= 0 ? 7u - (uint32_t)OCT : w.octinv), rest);
            if (rest)
            {
                if (st.sp < kPairs)
                    st.lds[(2 * st.sp) * kBlock] = c.g_base, st.lds[(2 * st.sp + 1) * kBlock] = c.g_mask;
                else
                    st.spill[st.sp - kPairs] = make_uint2(c.g_base, c.g_mask);
                ++st.sp;
            }
            src = bvh.nodes8 + (kWideNodeStride / 4u) * (size_t)node;
        }
        WideNode nd;
        nd.h0 = src[0], nd.h1 = src[1], nd.q2 = src[2];
        if (tri_lane)
        {
            // (an occlusion test reads 48 of the record's 64 bytes: the triangle id in the fourth piece is the closest hit's business)
            if (tri_occludes(r, nd.h0, nd.h1, nd.q2)) return true;
        }
        else
        {
            nd.q3 = src[3], nd.q4 = src[4];
            wide_node_test<kAnyOrder, OCT>(nd, w, r.tmin, r.tmax, c);
        }
        // nothing due: the next node group off the stack, or done
        if (c.t_hits == 0u && (c.g_mask >> 24) == 0u)
        {
            if (st.sp == 0) return false;
            --st.sp;
            if (st.sp < kPairs)
                c.g_base = st.lds[(2 * st.sp) * kBlock], c.g_mask = st.lds[(2 * st.sp + 1) * kBlock];
            else
            {
                const uint2 e = st.spill[st.sp - kPairs];
                c.g_base = e.x, c.g_mask = e.y;
            }
        }
    }
}
}  // namespace cap
