// trace8.hip — closest hit for the extension-ray queue (rt_indirect.hlsl:173; the driver's TraceRay in the reference) on the
// compressed 8-wide view of the tree (cap_wide.h), gfx950.
//
// One ray per lane, persistent waves.  What this kernel is built around (profiles/r01_tree_path.txt: the 4-wide kernel kept the
// CU's texture-address path busy 64 % of the launch and the vector ALU 38 %, with 11..35 % of the lanes doing useful work):
//   * bytes per lane and step: 80 B for eight child boxes (five 16-B loads) instead of 112 B for four;
//   * one stack entry per visited node (the rest of its node group, 8 B) instead of up to three: the LDS part of the stack is 8
//     entries x 8 B per lane (16 KB per workgroup), deeper entries go to the thread's spill slice;
//   * lane refill from a per-wave ray buffer in LDS: a wave's next 64-ray chunk is loaded into registers one chunk ahead (its
//     chunk index comes from the class's work counter, grabbed one chunk ahead as in the fused kernels) and parked in LDS when
//     the previous chunk has been handed out, so a lane that retires takes its next ray with two ds_reads instead of making the
//     whole wave wait for a global round trip -- which is what allows refilling at 16 idle lanes instead of 28;
//   * while-while: per iteration the wave runs either the node step (lanes with a node due) or the triangle step, whichever has
//     enough lanes; the other lanes wait, so an iteration pays for one of the two bodies.
// The hit rule (minimum t, ties to the lower triangle id; DESIGN.md "Intersection contract") is visit-order independent and the
// boxes are conservative (wide_builder.cpp), so the records written are bit-identical to every other traversal of the build.
#include "cap_kernels.h"
#include "cap_wide_trace.h"

namespace cap
{
#ifndef CAP_W8_LDS
#define CAP_W8_LDS 8  // LDS stack entries (8 B) per lane
#endif
#ifndef CAP_W8_BLOCKS
#define CAP_W8_BLOCKS 6  // workgroups per CU the kernel is register-allocated for
#endif
#ifndef CAP_W8_REFILL
#define CAP_W8_REFILL 16  // refill once this many lanes are idle
#endif
constexpr int      kW8Lds       = CAP_W8_LDS;
// pair-stack capacity the host checks the wide tree's depth against (LDS part + the thread's spill slice)
constexpr uint32_t kW8StackPairs = (uint32_t)kW8Lds + kSpillEntries / 2u;

// Diagnostic build only (make EXTRA=-DCAP_W8_COUNT): lane- and wave-level step counts of the closest-hit kernel, read by
// tools/w8_counts.py: [0] node steps (lanes), [1] triangle tests (lanes), [2] load sequences (wave iterations with live lanes),
// [3] node steps on the first kWideTopNodes nodes, [4] rays, [5] loop iterations (waves), [6] stack pushes, [7] spilled pushes,
// [8] node steps whose node's own entry distance (as its parent's test computed it) was already beyond best_t when it was picked (what a
// per-child cull would skip), [9] of those, the ones a per-GROUP bound (minimum over the children still in the group) would skip,
// [10] node steps that hit no child at all
#ifdef CAP_W8_COUNT
__device__ unsigned long long g_w8_counts[16];
#define W8_COUNT(i, v) (cnt[i] += (v))
#else
#define W8_COUNT(i, v) ((void)0)
#endif

// CAMERA: the queue is the camera rays' identity queue (entry i = frame slot i / Ppad, local pixel i % Ppad; class k owns entries
// [k * capacity, min((k + 1) * capacity, total)): dense scenes, context.hip primary_wide) and is never materialised -- a chunk's rays are
// generated where the other instantiation loads them (the same primary_dir() on the same operands as k_raygen_identity, which this
// replaces: 32 B written and read per camera ray and one launch less; docs/experiments.md (89)).
struct CameraFeed
{
    CameraDev         cam;
    ScreenDev         screen;
    const FrameConst* frames;
    uint32_t          n_slots;
};
template <bool CAMERA>
__global__ __launch_bounds__(kBlock, CAP_W8_BLOCKS) void k_trace_closest8(BvhDev bvh, RayQueue q, float4* hits, uint32_t* work, uint32_t refill_idle,
                                                                          CameraFeed feed)
{
    __shared__ uint2  lds_stack[kW8Lds * kBlock];
    __shared__ float4 lds_rays[2 * kBlock];  // per wave: 64 x (origin, tmin) then 64 x (direction, tmax)
    const uint32_t lane = threadIdx.x & 63u;
    float4* const  rbuf = lds_rays + (threadIdx.x >> 6) * 128u;
    const uint32_t my_class = wave_global_id() % kQueueClasses;
    uint32_t       n_class;
    if (CAMERA)
    {
        const uint64_t total = (uint64_t)feed.n_slots * feed.screen.pixels_padded, begin = (uint64_t)my_class * q.class_capacity;
        n_class              = begin < total ? (uint32_t)(total - begin < q.class_capacity ? total - begin : q.class_capacity) : 0u;
    }
    else
    {
        n_class = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.count[my_class * kCounterStride]);
        n_class = n_class < q.class_capacity ? n_class : q.class_capacity;
    }

    // ---- feed: chunk j of this wave's class, one chunk ahead in registers (pa, pb), the current one in LDS (rbuf) ----
    uint32_t grab   = grab_issue(work, my_class);
    uint32_t pend_n = 0, pend_base = 0, buf_n = 0, buf_pos = 0, buf_base = 0;
    float4   pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
    auto     fetch = [&]() {  // wave-uniform
        const uint32_t start = grab_value(grab) * 64u;
        pend_n               = 0;
        if (start >= n_class) return;  // past the end of this class's sub-queue: the feed has ended
        pend_n    = n_class - start < 64u ? n_class - start : 64u;
        pend_base = my_class * q.class_capacity + start;
        if (CAMERA)
        {
            if (lane < pend_n)
            {
                const uint32_t i = pend_base + lane, slot = i / feed.screen.pixels_padded, pl = i - slot * feed.screen.pixels_padded;
                uint32_t       x = 0, y = 0;
                const bool     in_image = local_pixel_to_xy(feed.screen, pl, x, y);  // (padding lanes of partial tiles: an empty interval)
                const v3       d        = in_image ? primary_dir(feed.cam, feed.screen, feed.frames[slot], x, y) : mk3(0.f, 0.f, 1.f);
                pa = make_float4(feed.cam.position[0], feed.cam.position[1], feed.cam.position[2], 0.0f);
                pb = make_float4(d.x, d.y, d.z, in_image ? kPrimaryFar : 0.0f);
            }
        }
        else if (lane < pend_n)
            pa = q.org_tmin[pend_base + lane], pb = q.dir_tmax[pend_base + lane];
        grab = grab_issue(work, my_class);
    };
    fetch();

    WideStack<kW8Lds> st{lds_stack + threadIdx.x, wide_spill_of_thread(bvh), 0};
    bool              alive = false;
    Ray               r     = make_ray(mk3(0, 0, 0), mk3(0, 0, 1), 0.f, 0.f);
    WideRay           w     = make_wide_ray(r.o, r.d);
    WideCursor        c;
    wide_cursor_root(c);
    float    best_t = 0.f, best_u = 0.f, best_v = 0.f;
    uint32_t best_gid = kInvalidId, out = 0;
#ifdef CAP_W8_COUNT
    unsigned long long cnt[16] = {};
    float cur_tn[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // entry distances of the current node group's children, by slot
    float tn_stack[24][8];
#endif
    while (true)
    {
        W8_COUNT(5, lane == 0 ? 1 : 0);
        unsigned long long m_alive = __ballot(alive);
        if (64u - (uint32_t)__popcll(m_alive) >= refill_idle)
        {
            // hand rays to idle lanes: the rest of the parked chunk, then (once) the start of the next one
            for (int rep = 0; rep < 2; ++rep)
            {
                if (buf_pos >= buf_n)
                {
                    if (pend_n == 0) break;
                    if (lane < pend_n) rbuf[lane] = pa, rbuf[64u + lane] = pb;
                    buf_n = pend_n, buf_pos = 0, buf_base = pend_base;
                    fetch();
                    wave_handoff();
                }
                const unsigned long long idle = ~m_alive;
                const uint32_t n_idle = (uint32_t)__popcll(idle), avail = buf_n - buf_pos;
                const uint32_t take   = avail < n_idle ? avail : n_idle;
                const uint32_t rank   = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (!alive && rank < take)
                {
                    const uint32_t e = buf_pos + rank;
                    const float4   a = rbuf[e], b = rbuf[64u + e];
                    r      = make_ray(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w, b.w);
                    w      = make_wide_ray(r.o, r.d);
                    best_t = r.tmax, best_u = 0.f, best_v = 0.f, best_gid = kInvalidId;
                    out    = buf_base + e;
                    wide_cursor_root(c);
                    st.sp = 0, alive = true;
                    W8_COUNT(4, 1);
#ifdef CAP_W8_COUNT
                    for (int k = 0; k < 8; ++k) cur_tn[k] = 0.f;
#endif
                }
                buf_pos += take;
                m_alive = __ballot(alive);
                if (m_alive == ~0ull) break;
            }
        }
        if (m_alive == 0ull)
        {
            if (buf_pos >= buf_n && pend_n == 0) break;  // feed ended and every lane retired
            continue;
        }
        // invariant: an alive lane has a triangle or a node due.  ONE load sequence serves both kinds of lane: a divergent 16-B
        // load costs the CU's texture-address path the same ~29 cycles whether 10 or 64 lanes take part (it works in quads of
        // lanes, and the lanes of either kind are scattered over all quads) (measured: three extra 16-B loads per node step
        // cost 1.8 ms; serving 45 % of the node steps from an LDS copy of the top levels, lane by lane, saved nothing).  So a
        // lane with a triangle due fetches its 64-B record with the first four of the five loads a node lane needs, and both
        // tests run on what arrived.  (Round 5, docs/experiments.md (59)(60): that path is the SECOND limiter; the first is the
        // vector ALU at the real cost of this loop's instruction classes, ~1 260 SIMD cycles per iteration, and the memory
        // system behind both delivers twice what the kernel asks of it.)
        const bool     tri_lane = alive && c.t_hits != 0u, node_lane = alive && c.t_hits == 0u;
        const float4*  src      = bvh.nodes8;
        if (tri_lane)
        {
            src = bvh.tris8 + 4 * (size_t)wide_pick_triangle(c);
            W8_COUNT(1, 1);
        }
        else if (node_lane)
        {
            bool           rest;
#ifdef CAP_W8_COUNT
            {
                const uint32_t bit = 31u - (uint32_t)__clz((int)c.g_mask), slot = (bit - 24u) ^ w.octinv;
                float          gmin = 3e38f;
                for (uint32_t b = 24u; b < 32u; ++b)
                    if ((c.g_mask >> b) & 1u) gmin = fminf(gmin, cur_tn[(b - 24u) ^ w.octinv]);
                if (cur_tn[slot] > best_t) cnt[8] += 1;
                if (gmin > best_t) cnt[9] += 1;
            }
#endif
            const uint32_t node = wide_pick_child(c, w.octinv, rest);
            W8_COUNT(0, 1);
            W8_COUNT(6, rest ? 1 : 0);
            W8_COUNT(7, (rest && st.sp >= kW8Lds) ? 1 : 0);
            W8_COUNT(3, node < kWideTopNodes ? 1 : 0);
#ifdef CAP_W8_COUNT
            if (rest && st.sp < 24)
                for (int k = 0; k < 8; ++k) tn_stack[st.sp][k] = cur_tn[k];
#endif
            if (rest) st.push(c.g_base, c.g_mask);
            src = bvh.nodes8 + (kWideNodeStride / 4u) * (size_t)node;
        }
        W8_COUNT(2, lane == 0 ? 1 : 0);
        WideNode nd;
        // "defined, whatever it holds": a lane that does not load never reads these, so no instruction is spent on a value for it
        // (zero-initialising the 20 registers was 20 moves per iteration; leaving the struct uninitialised made the allocator spill)
#define CAP_DEF4(v) asm volatile("" : "=v"((v).x), "=v"((v).y), "=v"((v).z), "=v"((v).w))
        CAP_DEF4(nd.h0);
        CAP_DEF4(nd.h1);
        CAP_DEF4(nd.q2);
        CAP_DEF4(nd.q3);
        CAP_DEF4(nd.q4);
#undef CAP_DEF4
        if (alive) nd.h0 = src[0], nd.h1 = src[1], nd.q2 = src[2], nd.q3 = src[3];
        if (node_lane) nd.q4 = src[4];
        if (tri_lane)
        {
            float t, u, v;
            if (tri_test(r, nd.h0, nd.h1, nd.q2, t, u, v))
            {
                const uint32_t gid = f2u(nd.q3.x);
                if (t < best_t || (t == best_t && gid < best_gid)) best_t = t, best_u = u, best_v = v, best_gid = gid;
            }
        }
#ifdef CAP_W8_COUNT
        if (node_lane)
        {
            wide_node_test(nd, w, r.tmin, best_t, c, cur_tn);
            if (c.t_hits == 0u && (c.g_mask >> 24) == 0u) cnt[10] += 1;
        }
#else
        if (node_lane) wide_node_test(nd, w, r.tmin, best_t, c);
#endif
        // a lane with nothing due takes the next node group off its stack, or retires
        if (alive && c.t_hits == 0u && (c.g_mask >> 24) == 0u)
        {
            if (st.sp == 0)
            {
                hits[out] = make_float4(best_u, best_v, u2f(best_gid), best_t);
                alive     = false;
            }
            else
            {
                st.pop(c);
#ifdef CAP_W8_COUNT
                if (st.sp < 24)
                    for (int k = 0; k < 8; ++k) cur_tn[k] = tn_stack[st.sp][k];
#endif
            }
        }
    }
#ifdef CAP_W8_COUNT
    for (int i = 0; i < 16; ++i)
    {
        unsigned long long v = cnt[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0 && v) atomicAdd(&g_w8_counts[i], v);
    }
#endif
}

// Shadow rays of the reference model on the wide view with the closest-hit kernel's lane refill (dense scenes only, see
// launch_trace_any8_refill): a lane that is done -- occluded at its first hit, or through the tree without one -- takes the next entry of
// its class from the wave's LDS buffer.  The per-chunk kernel (k_trace_any<24>) keeps a wave on one chunk until its LONGEST ray is done,
// with 29 of 64 lanes active on average; where the tree is cache-resident that costs nothing (a load sequence is paid per lane:
// docs/experiments.md (44)), but where a step's round trip ends in HBM an iteration lasts 4 us whatever the number of lanes in it, and
// lanes are throughput.  Same tests, same additions to the same plane entries by their only writers: bit-identical.
// Entry format (reference model): (origin, path id) (contribution, -); direction = the light of the path's frame, tmin / tmax constants.
__global__ __launch_bounds__(kBlock, CAP_W8_BLOCKS) void k_trace_any8_refill(BvhDev bvh, ShadowQueue q, float4* target, uint32_t pixels_padded,
                                                                             uint32_t n_slots, uint64_t* guard, uint32_t* work, const FrameConst* frames,
                                                                             uint32_t refill_idle)
{
    __shared__ uint2  lds_stack[kW8Lds * kBlock];
    __shared__ float4 lds_rays[kBlock];  // per wave: 64 x (origin, path id)
    __shared__ float4 lds_light[kMaxFrameSlots];
    for (uint32_t k = threadIdx.x; k < n_slots && k < kMaxFrameSlots; k += kBlock)
        lds_light[k] = make_float4(frames[k].light_dir[0], frames[k].light_dir[1], frames[k].light_dir[2], 0.f);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    float4* const  rbuf = lds_rays + (threadIdx.x >> 6) * 64u;
    const uint32_t my_class = wave_global_id() % kQueueClasses;
    uint32_t       n_class  = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.count[my_class * kCounterStride]);
    n_class                 = n_class < q.class_capacity ? n_class : q.class_capacity;

    uint32_t grab   = grab_issue(work, my_class);
    uint32_t pend_n = 0, pend_base = 0, buf_n = 0, buf_pos = 0, buf_base = 0;
    float4   pa = make_float4(0.f, 0.f, 0.f, 0.f);
    auto     fetch = [&]() {  // wave-uniform
        const uint32_t start = grab_value(grab) * 64u;
        pend_n               = 0;
        if (start >= n_class) return;
        pend_n    = n_class - start < 64u ? n_class - start : 64u;
        pend_base = my_class * q.class_capacity + start;
        if (lane < pend_n) pa = q.org_tmin[pend_base + lane];
        grab = grab_issue(work, my_class);
    };
    fetch();

    WideStack<kW8Lds> st{lds_stack + threadIdx.x, wide_spill_of_thread(bvh), 0};
    bool              alive = false;
    Ray               r     = make_ray(mk3(0, 0, 0), mk3(0, 0, 1), 0.f, 0.f);
    WideRay           w     = make_wide_ray(r.o, r.d);
    WideCursor        c;
    wide_cursor_root(c);
    uint32_t entry = 0;
    size_t   idx   = 0;
    bool     good  = false;
    while (true)
    {
        unsigned long long m_alive = __ballot(alive);
        if (64u - (uint32_t)__popcll(m_alive) >= refill_idle)
        {
            for (int rep = 0; rep < 2; ++rep)
            {
                if (buf_pos >= buf_n)
                {
                    if (pend_n == 0) break;
                    if (lane < pend_n) rbuf[lane] = pa;
                    buf_n = pend_n, buf_pos = 0, buf_base = pend_base;
                    fetch();
                    wave_handoff();
                }
                const unsigned long long idle = ~m_alive;
                const uint32_t n_idle = (uint32_t)__popcll(idle), avail = buf_n - buf_pos;
                const uint32_t take   = avail < n_idle ? avail : n_idle;
                const uint32_t rank   = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (!alive && rank < take)
                {
                    const uint32_t e   = buf_pos + rank;
                    const float4   a   = rbuf[e];
                    const uint32_t pid = f2u(a.w);
                    entry = buf_base + e;
                    good  = (pid >> kPidShift) < n_slots && (pid & kPidMask) < pixels_padded;
                    idx   = (size_t)(pid >> kPidShift) * pixels_padded + (pid & kPidMask);
                    if (!good)
                    {
                        // never true for a well-formed queue; reported through CapStats::guard_* instead of faulting
                        atomicAdd((unsigned long long*)guard + 2, 1ull);
                        guard[3] = ((uint64_t)entry << 32) | pid;
                    }
                    const float4 L = lds_light[good ? (pid >> kPidShift) : 0u];
                    r = make_ray(mk3(a.x, a.y, a.z), mk3(L.x, L.y, L.z), kRayEps, good ? kRayFar : 0.0f);  // malformed entry: empty interval
                    w = make_wide_ray(r.o, r.d);
                    wide_cursor_root(c);
                    st.sp = 0, alive = true;
                }
                buf_pos += take;
                m_alive = __ballot(alive);
                if (m_alive == ~0ull) break;
            }
        }
        if (m_alive == 0ull)
        {
            if (buf_pos >= buf_n && pend_n == 0) break;  // feed ended and every lane retired
            continue;
        }
        const bool    tri_lane = alive && c.t_hits != 0u, node_lane = alive && c.t_hits == 0u;
        const float4* src      = bvh.nodes8;
        if (tri_lane)
            src = bvh.tris8 + 4 * (size_t)wide_pick_triangle(c);
        else if (node_lane)
        {
            bool           rest;
            const uint32_t node = wide_pick_child(c, kAnyOct(w.octinv), rest);
            if (rest) st.push(c.g_base, c.g_mask);
            src = bvh.nodes8 + (kWideNodeStride / 4u) * (size_t)node;
        }
        WideNode nd;
#define CAP_DEF4(v) asm volatile("" : "=v"((v).x), "=v"((v).y), "=v"((v).z), "=v"((v).w))  // (see k_trace_closest8)
        CAP_DEF4(nd.h0);
        CAP_DEF4(nd.h1);
        CAP_DEF4(nd.q2);
        CAP_DEF4(nd.q3);
        CAP_DEF4(nd.q4);
#undef CAP_DEF4
        if (alive) nd.h0 = src[0], nd.h1 = src[1], nd.q2 = src[2];
        if (node_lane) nd.q3 = src[3], nd.q4 = src[4];  // (a triangle lane needs 48 of its record's 64 bytes: no id for an occlusion test)
        bool occluded = false;
        if (tri_lane) occluded = tri_occludes(r, nd.h0, nd.h1, nd.q2);
        if (node_lane) wide_node_test<kAnyOrder>(nd, w, r.tmin, r.tmax, c);
        if (occluded)
            alive = false;  // lighting.h:57: an occluded ray adds nothing
        else if (alive && c.t_hits == 0u && (c.g_mask >> 24) == 0u)
        {
            if (st.sp == 0)
            {
                // through the tree without an occluder: the contribution evaluated at shading time is added (lighting.h:57-60)
                if (good)
                {
                    const float4 con = q.contrib_pid[entry], cur = target[idx];
                    target[idx]      = make_float4(cur.x + con.x, cur.y + con.y, cur.z + con.z, cur.w);
                }
                alive = false;
            }
            else
                st.pop(c);
        }
    }
}

// Idle lanes at which a wave refills (A/B switch CAP_W8_REFILL), clamped to 1..64: above 64 the refill condition is never true, no
// lane ever gets a ray and the persistent loop would spin for ever.
static uint32_t w8_refill_idle(const LaunchCfg& cfg)
{
    const long v = (long)cfg.sw_get(SW_W8_REFILL, CAP_W8_REFILL);
    return (uint32_t)(v < 1 ? 1 : v > 64 ? 64 : v);
}

void launch_trace_any8_refill(const LaunchCfg& cfg, const BvhDev& bvh, const ShadowQueue& q, uint32_t max_count, float4* target, uint32_t pixels_padded,
                              uint32_t n_slots, uint64_t* guard, uint32_t* work, const FrameConst* frames)
{
    uint32_t g = (max_count + kBlock - 1) / kBlock;
    uint32_t cap = cfg.cu_count ? cfg.cu_count * (uint32_t)CAP_W8_BLOCKS : cfg.grid_blocks;
    if ((uint64_t)cap * kBlock > bvh.spill_threads) cap = bvh.spill_threads / kBlock;  // every thread owns a spill slice
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    const uint32_t refill = w8_refill_idle(cfg);
    hipLaunchKernelGGL(k_trace_any8_refill, dim3(g), dim3(kBlock), 0, cfg.stream, bvh, q, target, pixels_padded, n_slots, guard, work, frames, refill);
}

#ifdef CAP_W8_COUNT
extern "C" int cap_debug_w8_counts(unsigned long long* out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w8_counts), sizeof(g_w8_counts));
    if (e == hipSuccess && reset)
    {
        unsigned long long z[16] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_w8_counts), z, sizeof(z));
    }
    return (int)e;
}
#endif

// no geometry: every queued ray misses
__global__ __launch_bounds__(kBlock) void k_trace_closest8_empty(RayQueue q, float4* hits)
{
    const uint32_t slots = (q.class_capacity >> 6) * kQueueClasses;
    for (uint32_t cs = wave_global_id(); cs < slots; cs += wave_total())
    {
        uint32_t i, klass;
        if (queue_chunk(q.count, q.class_capacity, cs, threadIdx.x & 63u, i, klass)) hits[i] = make_float4(0.f, 0.f, u2f(kInvalidId), 0.f);
    }
}

uint32_t wide8_stack_pairs() { return kW8StackPairs; }

static void launch_closest8(const LaunchCfg& cfg, const BvhDev& bvh, const RayQueue& q, uint32_t max_count, float4* hits, uint32_t* work,
                            const CameraFeed* camera)
{
    uint32_t g = (max_count + kBlock - 1) / kBlock;
    const uint32_t per_cu = (uint32_t)cfg.sw_get(SW_W8_GRID, CAP_W8_BLOCKS);  // A/B switch
    uint32_t cap = cfg.cu_count ? cfg.cu_count * per_cu : cfg.grid_blocks;
    if ((uint64_t)cap * kBlock > bvh.spill_threads) cap = bvh.spill_threads / kBlock;  // every thread owns a spill slice
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    if (bvh.tri_count == 0)
    {
        hipLaunchKernelGGL(k_trace_closest8_empty, dim3(g), dim3(kBlock), 0, cfg.stream, q, hits);
        return;
    }
    const uint32_t refill = w8_refill_idle(cfg);
    if (camera)
        hipLaunchKernelGGL(k_trace_closest8<true>, dim3(g), dim3(kBlock), 0, cfg.stream, bvh, q, hits, work, refill, *camera);
    else
        hipLaunchKernelGGL(k_trace_closest8<false>, dim3(g), dim3(kBlock), 0, cfg.stream, bvh, q, hits, work, refill, CameraFeed{});
}

void launch_trace_closest8(const LaunchCfg& cfg, const BvhDev& bvh, const RayQueue& q, uint32_t max_count, float4* hits, uint32_t* work)
{
    launch_closest8(cfg, bvh, q, max_count, hits, work, nullptr);
}

// Camera rays of a dense scene: q carries only the identity queue's class capacity (its planes and counters are not touched).
void launch_trace_closest8_camera(const LaunchCfg& cfg, const BvhDev& bvh, const RayQueue& q, uint32_t max_count, float4* hits, uint32_t* work,
                                  const CameraDev& cam, const ScreenDev& screen, const FrameConst* frames, uint32_t n_slots)
{
    const CameraFeed feed{cam, screen, frames, n_slots};
    launch_closest8(cfg, bvh, q, max_count, hits, work, &feed);
}
}  // namespace cap
