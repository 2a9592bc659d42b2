// context.hip — implementation of the C ABI in include/capsaicin_hip.h: device memory ownership, uploads, the
// per-batch wavefront loop (the MI355X counterpart of RaytracingSystem::Run, reference
// src/systems/raytracing_system.cpp:230-318, ray passes only), readback, statistics.
//
// The product path is HIP only: every entry point that needs the GPU fails with CAP_ERR_HIP when no device or
// kernel is available; there is no CPU fallback.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/capsaicin_hip.h"
#include <chrono>

#include "cap_kernels.h"
#include "sah_builder.h"
#include "wide_builder.h"
#include "cap_wide.h"

using namespace cap;

namespace
{
thread_local std::string g_error;

int fail(int code, const char* fmt, ...)
{
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                         \
    do                                                                                                        \
    {                                                                                                         \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return fail(CAP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                                \
    } while (0)

template <typename T>
struct DevBuf
{
    T*     p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr, o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept
    {
        if (this != &o)
        {
            release();
            p = o.p, n = o.n, o.p = nullptr, o.n = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr, n = 0;
    }
    hipError_t ensure(size_t count)
    {
        if (count <= n && p) return hipSuccess;
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
};

// camera rays take the per-lane wide kernel instead of the packet walk from this many triangles per pixel on (measured: cap_render;
// 1.0 until the walk's node step was rewritten in round 6 -- at 1.0 the walk now wins by 6 % of the step, at 2.0 the two are level)
constexpr double kPrimaryWideTrianglesPerPixel = 2.0;
// shadow rays take the lane-refill kernel from this many bytes of wide nodes + intersection records on (measured: cap_render).
// Round 6: never by default.  Until the per-chunk kernel's traversal was compiled per light octant (docs/experiments.md (86)) the refill
// kernel won by 3 % from 512 MiB of tree on; now the per-chunk kernel wins by 6 - 8 % at 8.4 M and 16.8 M triangles (1.5 GB of tree),
// as it always did below.  CAP_ANY_REFILL=1 still selects the refill kernel (tests/test_fallback_kernels_gpu.py keeps it honest).
constexpr uint64_t kAnyRefillTreeBytes = ~0ull;
// AUTO builds with surface-area splits (ploc.hip, sah_device) from this many triangles on, the clustering alone below (see cap_bvh_build)
constexpr uint32_t kAutoSahTriangles = 4096;

// guard block of a context (ShadeArgs::shaded_counter): {-, malformed path ids seen by shade, by trace_any, last offender, appends
// beyond a class's capacity, -, -, -}
constexpr size_t kGuardWords = 8;

enum StageId
{
    ST_PRIMARY,
    ST_CLOSEST,
    ST_ANY,
    ST_SHADE,
    ST_RESOLVE,
    ST_POST,
    ST_COUNT,
    ST_DIRECT,      // "RT Direct lighting": second label of the bounce-0 shading / shadow-ray spans
    ST_POST_PASS0,  // .. ST_POST_PASS0 + 4: "Spatial gather", "Temporal upscale", "EAW", "Combine illumination", "TAA"
    ST_NONE = -1
};

struct TimedSpan
{
    hipEvent_t a, b;
    int        stage;
    int        also = ST_NONE;  // a span may count under a second label
};
}  // namespace

struct CapContext
{
    int         device = 0;
    hipStream_t stream = nullptr;
    bool        own_stream = false;
    int         cu_count = 256;

    // scene (GeometryStorage layout, asset_load_system.h:16-27)
    DevBuf<float>    positions, normals, texcoords;
    DevBuf<uint32_t> indices;
    DevBuf<uint4>    tri_ids;
    DevBuf<uint4>    mesh_offsets;
    DevBuf<uint32_t> mesh_texture;
    uint32_t         vertex_count = 0, index_count = 0, mesh_count = 0, tri_count = 0;
    bool             scene_ready = false;

    std::vector<DevBuf<uint8_t>> texture_data;
    std::vector<TextureDev>      texture_host;
    DevBuf<TextureDev>           textures;
    bool                         textures_dirty = true;
    DevBuf<float2>               bluenoise;
    bool                         bluenoise_ready = false;
    std::vector<CapMaterial>     materials_host;
    DevBuf<CapMaterial>          materials;
    // EXT: host copies needed to build the light table, the table itself, the B/A blue-noise channels
    std::vector<float>           positions_host;
    std::vector<uint32_t>        indices_host;
    std::vector<CapMeshDesc>     meshes_host;
    DevBuf<uint32_t>             light_tris;
    DevBuf<float>                light_cdf;
    DevBuf<float2>               bluenoise_ba;
    uint32_t                     light_count = 0;
    float                        light_area  = 0.0f;
    bool                         materials_ready = false;

    // BVH
    DevBuf<float4>   shade_tris, tris_sorted, nodes, tri_raw, tri_box;
    DevBuf<float4>   nodes8, tris8;           // compressed 8-wide view (cap_wide.h) and its intersection records
    DevBuf<uint32_t> wide_src;                // leaf-order index per wide-order record
    DevBuf<uint32_t> wide_task, wide_alloc, wide_cnt;   // device collapse: binary node per wide node, allocation counters, per-level bases
    uint32_t         wide8_depth = 0, wide8_top = 0, wide8_nodes = 0;
    float            wide8_ms = 0.f;
    DevBuf<uint32_t> stack_spill;             // traversal-stack entries beyond the LDS part, per thread of the persistent grid
    DevBuf<float4>   fan_pairs, fan_singles;  // exhaustive path: fan-pair records (5 float4) and the unpaired triangles (4 float4)
    uint32_t         fan_pair_count = 0, fan_single_count = 0;
    DevBuf<float4>   fan_pairs_nee;           // the pair records again, potential occluders of next-event rays first (update_nee_pairs)
    uint32_t         fan_pair_nee_count = 0;
    std::vector<float>    fan_pairs_host;     // 20 floats per pair (+ padding records), as uploaded
    std::vector<uint32_t> light_tris_host;    // global ids of the emissive triangles (cap_materials_upload)
    DevBuf<uint32_t> leaf_tri, keys0, keys1, vals0, vals1, hist, parent, flags, bvh_misc;  // bvh_misc: 6 bounds + depth
    DevBuf<float4>   ploc_boxes;  // CAP_BVH_BUILD_PLOC scratch (ploc.hip)
    DevBuf<uint32_t> ploc_ints;
    DevBuf<uint32_t> sahdev_words;  // CAP_BVH_BUILD_SAH_DEVICE scratch (ploc.hip)
    CapBvhInfo       bvh_info{};
    bool             bvh_ready = false;

    // camera / screen
    CapCameraData camera{}, prev_camera{};
    bool          camera_ready = false, prev_camera_ready = false;
    ScreenDev     screen{};
    uint64_t      max_batch_paths = 0;
    SwitchTable   sw{};                    // A/B switches: environment at creation, then cap_debug_set(CAP_DEBUG_SWITCH_BASE + i)
    uint32_t      debug_capacity_div = 1;  // cap_debug_set(CAP_DEBUG_QUEUE_CAPACITY_DIV): tests of the append guard only
    uint32_t      debug_wide_depth_limit = 0;  // cap_debug_set(CAP_DEBUG_WIDE_DEPTH_LIMIT): pretend the wide kernels' stacks end here
    bool          debug_fail_lane1 = false;    // cap_debug_set(CAP_DEBUG_FAIL_LANE1): the second working set "cannot be allocated"
    // The smallest second working set (paths = slots x padded pixels) whose allocation has failed since the last call that can free
    // memory: cap_render does not try that size or a larger one again (every call repeated ~28 GB of hipMalloc / hipFree).
    uint64_t      lane1_failed_paths = 0;
    uint32_t      lane1_failed_bounces = 0, lane1_retry_tick = 0;
    uint32_t      lanes_last_render = 0;       // cap_debug_get(CAP_DEBUG_LANES_USED)
    uint32_t      last_class_capacity = 0;     // sub-queue capacity of the last cap_render batch (CAP_DEBUG_QUEUE_CANARY_*)
    uint32_t      traversal_mode  = CAP_TRAVERSAL_AUTO;
    uint32_t      bvh_build_mode  = CAP_BVH_BUILD_AUTO;

    // wavefront state
    DevBuf<float4>     hits, q_org[2], q_dir[2], q_thr[2], s_org, s_dir, s_con, pl_color, pl_direct, pl_albedo, aov_geo, aov_nd,
        accum, image_tmp;
    DevBuf<uint32_t>   counters;  // per batch: ext[0..D], shadow[0..D]
    DevBuf<uint64_t>   shaded_counter;
    // Second working set for batches in flight beside the first (cap_render "two lanes"): queues, planes, counters and traversal
    // spill slices of its own, on a stream of its own.  Lane 0 is the set above on `stream`; the LAST batch of a call always runs
    // there, so everything that reads a finished frame's planes (AOV read-backs, the reconstruction chain) finds them where it did.
    struct Lane
    {
        DevBuf<float4>   hits, q_org[2], q_dir[2], q_thr[2], s_org, s_dir, s_con, pl_color, pl_direct, pl_albedo;
        DevBuf<uint32_t> counters, stack_spill;
    } lane1;
    hipStream_t stream2 = nullptr;
    hipEvent_t  lane_ev[2] = {nullptr, nullptr}, fork_ev = nullptr, join_ev = nullptr;
    // per-frame constants of a cap_render call: a ring of device buffers fed from pinned staging, so that a call need not wait
    // for the previous one (which may still be reading its own slot)
    static constexpr int kFrameRing = 4;
    DevBuf<FrameConst> frames_ring[kFrameRing];
    FrameConst*        frames_pinned[kFrameRing] = {nullptr, nullptr, nullptr, nullptr};
    size_t             frames_pinned_n[kFrameRing] = {0, 0, 0, 0};
    hipEvent_t         frames_event[kFrameRing] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t           frames_next = 0;
    uint32_t           slots_alloc = 0, bounces_alloc = 0;
    uint32_t           last_slots = 0;     // frame slots of the last batch rendered (AOV readback)
    bool               aov_valid = false;
    bool               aov_lowres = false;  // the AOV frame was rendered with CAP_RENDER_LOWRES_INDIRECT
    uint32_t           aov_frame  = 0;      // its frame_count (selects the 2x2 interleave offset)
    uint64_t           frames_accumulated = 0;

    // reconstruction chain (row-major W*H images)
    DevBuf<float4> post_in[4];  // indirect, direct, albedo, normal_depth of the frame
    DevBuf<float4> post_ihist[2], post_mhist[2], post_chist[2], post_prev_nd, post_itemp, post_temp[2], post_normals;
    uint32_t       post_w = 0, post_h = 0;
    int            post_last_dst = -1;

    // multi-GPU frame exchange (cap_comm_*): RCCL communicator, or `comm_local` when the shards of one process share a device
    void*           comm = nullptr;  // ncclComm_t
    uint32_t        comm_rank = 0, comm_size = 0;
    bool            comm_local = false;
    DevBuf<float>   comm_send, comm_gathered, comm_image;
    hipEvent_t      comm_event = nullptr;

    // statistics
    CapStats               stats{};
    std::vector<TimedSpan> spans;
    std::vector<std::array<hipEvent_t, 6>> post_marks;  // pass-boundary events of every cap_post_frame since the last sync
    std::vector<hipEvent_t> event_pool;
    struct PendingCounters
    {
        std::vector<uint32_t> host;  // filled by an async copy
        uint32_t              bounces;
    };
    std::vector<uint32_t*> pinned_pool;
    std::vector<std::pair<uint32_t*, uint32_t>> pending;  // (pinned counters, bounces) per batch, read at sync
    uint64_t*              pinned_shaded = nullptr;
    hipEvent_t             ev_begin = nullptr, ev_end = nullptr;
};

namespace
{
int sync_and_collect(CapContext* c)
{
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto& sp : c->spans)
    {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess)
        {
            for (int stage : {sp.stage, sp.also})
                switch (stage)
                {
                case ST_PRIMARY: c->stats.ms_primary += ms; break;
                case ST_CLOSEST: c->stats.ms_trace_closest += ms; break;
                case ST_ANY: c->stats.ms_trace_any += ms; break;
                case ST_SHADE: c->stats.ms_shade += ms; break;
                case ST_RESOLVE: c->stats.ms_resolve += ms; break;
                case ST_POST: c->stats.ms_post += ms; break;
                case ST_COUNT: c->stats.ms_total += ms; break;
                case ST_DIRECT: c->stats.ms_direct += ms; break;
                default:
                    if (stage >= ST_POST_PASS0 && stage < ST_POST_PASS0 + 5) c->stats.ms_post_pass[stage - ST_POST_PASS0] += ms;
                }
        }
        c->event_pool.push_back(sp.a);
        c->event_pool.push_back(sp.b);
    }
    c->spans.clear();
    for (auto& m : c->post_marks)
    {
        for (int pass = 0; pass < 5; ++pass)
        {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, m[pass], m[pass + 1]) == hipSuccess) c->stats.ms_post_pass[pass] += ms, c->stats.ms_post += ms;
        }
        for (hipEvent_t e : m) c->event_pool.push_back(e);
    }
    c->post_marks.clear();
    for (auto& pc : c->pending)
    {
        const uint32_t D = pc.second & 0xffffu;
        const bool     queued = (pc.second >> 16) == 0u;  // false: the batch's shadow rays were traced where they were generated
        const size_t per = (size_t)kQueueClasses * kCounterStride;
        for (uint32_t b = 0; b <= D; ++b)
            for (uint32_t k = 0; k < kQueueClasses; ++k)
            {
                c->stats.rays_extension += pc.first[b * per + k * kCounterStride];
                c->stats.rays_shadow += pc.first[b * per + k * kCounterStride + 1];
                if (queued) c->stats.shadow_entries += pc.first[b * per + k * kCounterStride + 1];
                if (b == 0)
                {
                    c->stats.rays_extension_bounce0 += pc.first[k * kCounterStride];
                    c->stats.rays_shadow_bounce0 += pc.first[k * kCounterStride + 1];
                    if (queued) c->stats.shadow_entries_bounce0 += pc.first[k * kCounterStride + 1];
                }
                // word 2: shadow rays the producer's probe answered (rays, though never queue entries); word 3: shaded vertices
                c->stats.rays_shadow += pc.first[b * per + k * kCounterStride + 2];
                if (b == 0) c->stats.rays_shadow_bounce0 += pc.first[k * kCounterStride + 2];
                c->stats.shaded_vertices += pc.first[b * per + k * kCounterStride + 3];
            }
        c->pinned_pool.push_back(pc.first);
    }
    c->pending.clear();
    if (c->pinned_shaded && c->shaded_counter.p)
    {
        HIP_TRY(hipMemcpy(c->pinned_shaded, c->shaded_counter.p, kGuardWords * sizeof(uint64_t), hipMemcpyDeviceToHost));
        c->stats.guard_shade     = c->pinned_shaded[1];
        c->stats.guard_trace_any = c->pinned_shaded[2];
        c->stats.guard_last      = c->pinned_shaded[3];
        c->stats.guard_append    = c->pinned_shaded[4];
    }
    return CAP_OK;
}

hipEvent_t get_event(CapContext* c)
{
    if (!c->event_pool.empty())
    {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct StageTimer
{
    CapContext* c;
    TimedSpan   sp;
    bool        on;
    StageTimer(CapContext* ctx, int stage, bool enabled = true, int also = ST_NONE) : c(ctx), on(enabled)
    {
        if (!on) return;
        sp.stage = stage;
        sp.also  = also;
        sp.a     = get_event(c);
        sp.b     = get_event(c);
        (void)hipEventRecord(sp.a, c->stream);
    }
    ~StageTimer()
    {
        if (!on) return;
        (void)hipEventRecord(sp.b, c->stream);
        c->spans.push_back(sp);
    }
};

void update_screen(CapContext* c, uint32_t w, uint32_t h, uint32_t shard_index, uint32_t shard_count)
{
    ScreenDev& s  = c->screen;
    s.width       = w;
    s.height      = h;
    s.tiles_x     = (w + kTileDim - 1) / kTileDim;
    s.tiles_y     = (h + kTileDim - 1) / kTileDim;
    s.tile_count  = s.tiles_x * s.tiles_y;
    s.shard_index = shard_index;
    s.shard_count = shard_count ? shard_count : 1;
    const uint32_t max_local = (s.tile_count + s.shard_count - 1) / s.shard_count;
    s.local_tiles   = s.tile_count > s.shard_index ? (s.tile_count - s.shard_index + s.shard_count - 1) / s.shard_count : 0;
    s.pixels_padded = max_local * kTilePixels;
    c->frames_accumulated = 0;
    c->aov_valid = false;
}

CameraDev camera_dev(const CapCameraData& cd)
{
    CameraDev d{};
    for (int k = 0; k < 3; ++k) d.position[k] = cd.position[k], d.right[k] = cd.right[k], d.forward[k] = cd.forward[k], d.up[k] = cd.up[k];
    d.focal_length = cd.focal_length;
    d.sensor_x     = cd.sensor_size[0];
    d.sensor_y     = cd.sensor_size[1];
    return d;
}

// lighting.h:20-33 + camera.h:41, evaluated once per frame on the host with the shared arithmetic contract
FrameConst frame_const(uint32_t frame_count, bool lowres_indirect)
{
    FrameConst f{};
    halton23(frame_count, f.jitter_x, f.jitter_y);
    f.frame_count = frame_count;
    // rt_indirect.hlsl:55-56: sp_offset = ((frame % 4) / 2, (frame % 4) % 2)
    f.lowres_sel = lowres_indirect ? (4u | (((frame_count % 4u) / 2u) << 1) | ((frame_count % 4u) % 2u)) : 0u;
    const float t = 2.0f * 3.14f * (float)(frame_count % 4096) / 4096.0f;
    float       st, ct;
    sincos_c(t, st, ct);
    const v3 dir = normalize3(mk3(40.0f * st, 100.0f, 40.0f * ct));
    f.light_dir[0] = dir.x, f.light_dir[1] = dir.y, f.light_dir[2] = dir.z;
    f.light_intensity[0] = 1.0f * (2.0f * 14.0f + 0.0f);
    f.light_intensity[1] = 1.0f * (2.0f * 12.0f + 0.0f);
    f.light_intensity[2] = 1.0f * (2.0f * 10.0f + (2.0f + 2.0f * ct));
    return f;
}

BvhDev bvh_dev(const CapContext* c)
{
    BvhDev b{};
    b.nodes     = c->nodes.p;
    b.tris      = c->tris_sorted.p;
    b.tris_by_id = c->tri_raw.p;
    b.stack_spill   = c->stack_spill.p;
    b.spill_threads = (uint32_t)(c->stack_spill.n / kSpillEntries);
    // A/B switch: the binary-tree kernels (also what runs if the 8-wide view's depth ever exceeds the pair stacks)
    const bool no_wide8 = c->sw.on(SW_NO_WIDE8);
    b.nodes8 = c->nodes8.p, b.tris8 = c->tris8.p;
    b.wide8_ok  = !no_wide8 && c->stack_spill.p && c->wide8_nodes != 0 && c->wide8_depth <= wide8_stack_pairs() + 1u &&
                 c->wide8_depth <= kWideLdsEntries / 2u + kSpillEntries / 2u + 1u &&
                 (c->debug_wide_depth_limit == 0u || c->wide8_depth <= c->debug_wide_depth_limit);
    b.wide8_top = c->wide8_top;
    b.fan_pairs = c->fan_pairs.p, b.fan_singles = c->fan_singles.p;
    b.fan_pair_count = c->fan_pair_count, b.fan_single_count = c->fan_single_count;
    b.fan_pairs_nee = c->fan_pairs_nee.p ? c->fan_pairs_nee.p : c->fan_pairs.p;
    b.fan_pair_nee_count = c->fan_pairs_nee.p ? c->fan_pair_nee_count : c->fan_pair_count;
    b.tri_count = c->tri_count;
    b.root      = c->tri_count >= 2 ? 0 : ~0;
    return b;
}

SceneDev scene_dev(const CapContext* c)
{
    SceneDev s{};
    s.shade_tris    = c->shade_tris.p;
    s.tri_ids       = c->tri_ids.p;
    s.mesh_texture  = c->mesh_texture.p;
    s.textures      = c->textures.p;
    s.texture_count = (uint32_t)c->texture_host.size();
    s.bluenoise     = c->bluenoise.p;
    s.kd_untextured = pow22_c(0.75f);
    s.bluenoise_ba  = c->bluenoise_ba.p;
    s.materials     = reinterpret_cast<const MaterialDev*>(c->materials.p);
    s.material_count = c->materials_ready ? c->mesh_count : 0u;
    s.light_tris    = c->light_tris.p;
    s.light_cdf     = c->light_cdf.p;
    s.light_count   = c->light_count;
    s.light_area    = c->light_area;
    return s;
}

int ensure_wavefront(CapContext* c, uint32_t slots, uint32_t bounces)
{
    const size_t planes_np = (size_t)slots * c->screen.pixels_padded;
    const size_t np        = planes_np + (size_t)kQueueClasses * 64;  // queue regions round up to whole chunks per class
    HIP_TRY(c->hits.ensure(np));
    for (int k = 0; k < 2; ++k)
    {
        HIP_TRY(c->q_org[k].ensure(np));
        HIP_TRY(c->q_dir[k].ensure(np));
        HIP_TRY(c->q_thr[k].ensure(np));
    }
    // the shadow queue's memory also holds the fused kernels' per-wave survivor rings (ShadeArgs::wave_ring): 128 entries for
    // every wave of the largest persistent grid (8 workgroups per CU)
    const size_t ring_np = (size_t)std::max(c->cu_count, 1) * 8 * (kBlock / 64) * 128;
    HIP_TRY(c->s_org.ensure(std::max(np, ring_np)));
    HIP_TRY(c->s_dir.ensure(np));
    HIP_TRY(c->s_con.ensure(std::max(np, ring_np)));
    HIP_TRY(c->pl_color.ensure(planes_np));
    HIP_TRY(c->pl_direct.ensure(planes_np));
    HIP_TRY(c->pl_albedo.ensure(planes_np));
    HIP_TRY(c->aov_geo.ensure(c->screen.pixels_padded));
    HIP_TRY(c->aov_nd.ensure(c->screen.pixels_padded));
    // queue counters (ext + shadow share words) + chunk-grab counters of the fused and of the any-hit launch, per bounce
    HIP_TRY(c->counters.ensure((3 * (size_t)(bounces + 1) + 1) * kQueueClasses * kCounterStride));  // + 1: the camera rays' identity queue
    if (!c->shaded_counter.p)
    {
        HIP_TRY(c->shaded_counter.ensure(kGuardWords));
        HIP_TRY(hipMemsetAsync(c->shaded_counter.p, 0, kGuardWords * sizeof(uint64_t), c->stream));
    }
    if (!c->accum.p || c->accum.n < c->screen.pixels_padded)
    {
        HIP_TRY(c->accum.ensure(c->screen.pixels_padded));
        HIP_TRY(hipMemsetAsync(c->accum.p, 0, sizeof(float4) * c->screen.pixels_padded, c->stream));
        c->frames_accumulated = 0;
    }
    return CAP_OK;
}

// the second lane's working set (same sizes as lane 0's), its stream and the events that order the two
int ensure_lane1(CapContext* c, uint32_t slots, uint32_t bounces)
{
    CapContext::Lane& L = c->lane1;
    const size_t planes_np = (size_t)slots * c->screen.pixels_padded;
    const size_t np        = planes_np + (size_t)kQueueClasses * 64;
    HIP_TRY(L.hits.ensure(np));
    for (int k = 0; k < 2; ++k)
    {
        HIP_TRY(L.q_org[k].ensure(np));
        HIP_TRY(L.q_dir[k].ensure(np));
        HIP_TRY(L.q_thr[k].ensure(np));
    }
    const size_t ring_np = (size_t)std::max(c->cu_count, 1) * 8 * (kBlock / 64) * 128;
    HIP_TRY(L.s_org.ensure(std::max(np, ring_np)));
    HIP_TRY(L.s_dir.ensure(np));
    HIP_TRY(L.s_con.ensure(std::max(np, ring_np)));
    HIP_TRY(L.pl_color.ensure(planes_np));
    HIP_TRY(L.pl_direct.ensure(planes_np));
    HIP_TRY(L.pl_albedo.ensure(planes_np));
    HIP_TRY(L.counters.ensure((3 * (size_t)(bounces + 1) + 1) * kQueueClasses * kCounterStride));
    if (c->stack_spill.n) HIP_TRY(L.stack_spill.ensure(c->stack_spill.n));
    if (!c->stream2)
    {
        // (A/B switch: the second lane's stream priority -- streams of different priority never share a hardware queue)
        if (c->sw.v[SW_LANE1_PRIORITY] >= 0)
            HIP_TRY(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, (int)c->sw.v[SW_LANE1_PRIORITY]));
        else
            HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    }
    for (hipEvent_t* e : {&c->lane_ev[0], &c->lane_ev[1], &c->fork_ev, &c->join_ev})
        if (!*e) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return CAP_OK;
}
}  // namespace

// (x, y) -> the four RGBA8 words a bilinear WRAP sample at that texel reads (cap_device.h TextureDev)
__global__ __launch_bounds__(256) void k_texture_footprints(const uint32_t* src, uint4* dst, uint32_t width, uint32_t height)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)width * height) return;
    const uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i - (size_t)y * width);
    const uint32_t x1 = x + 1 == width ? 0 : x + 1, y1 = y + 1 == height ? 0 : y + 1;
    dst[i] = make_uint4(src[(size_t)y * width + x], src[(size_t)y * width + x1], src[(size_t)y1 * width + x], src[(size_t)y1 * width + x1]);
}

extern "C" {

const char* cap_last_error(void) { return g_error.c_str(); }
// used by the host-side half of the library (obj_loader.cpp) so both report through cap_last_error()
void cap_set_error_(const char* msg) { g_error = msg ? msg : ""; }

int cap_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The switches' names: environment variables read ONCE, at cap_ctx_create (tools and profiling scripts set them around a whole process),
// and the strings cap_debug_switch_index() resolves for cap_debug_set.  A presence flag counts as set unless its value is "0".
static const char* const kSwitchNames[SW_COUNT] = {
    "CAP_NO_WIDE8", "CAP_LANE1_PRIORITY", "CAP_PLOC_RADIUS", "CAP_SAHDEV_LEAF", "CAP_WIDE_HOST_COLLAPSE", "CAP_TRACE_LAUNCHES", "CAP_NO_TWO_LANES",
    "CAP_LANE_SPLIT_MIN", "CAP_BLOCKS_PER_CU", "CAP_NO_CAMERA_CULL", "CAP_NO_ALBEDO_IN_W", "CAP_NO_INLINE_NEE", "CAP_NO_INLINE_PROBE", "CAP_NO_WAVE_RING",
    "CAP_ANY_REFILL", "CAP_PRIMARY_WIDE", "CAP_NO_PACKET", "CAP_NO_ANY_PROBE", "CAP_ANY_PROBE", "CAP_ANY_BLOCKS", "CAP_NO_PRIMARY_FUSE", "CAP_W8_REFILL",
    "CAP_W8_GRID", "CAP_AUTO_SAH_TRIANGLES", "CAP_NO_NEE_PAIR_CULL", "CAP_RAYGEN_KERNEL"};

static void switches_from_environment(SwitchTable& t)
{
    for (uint32_t k = 0; k < SW_COUNT; ++k)
    {
        const char* e = getenv(kSwitchNames[k]);
        t.v[k]        = -1;
        if (!e) continue;
        char*           end = nullptr;
        const long long v   = strtoll(e, &end, 10);
        t.v[k]              = (end != e && v >= 0) ? v : 1;  // "CAP_NO_X=" / "=yes": a set flag
    }
    if (getenv("CAP_BVH_BINARY") && t.v[SW_NO_WIDE8] < 0) t.v[SW_NO_WIDE8] = 1;  // older name of the same switch
}

int cap_debug_switch_index(const char* name)
{
    for (uint32_t k = 0; name && k < SW_COUNT; ++k)
        if (strcmp(name, kSwitchNames[k]) == 0) return (int)k;
    return -1;
}

int cap_ctx_create(int device_id, void* hip_stream, CapContext** out_ctx)
{
    if (!out_ctx) return fail(CAP_ERR_INVALID_ARG, "cap_ctx_create: out_ctx is NULL");
    *out_ctx = nullptr;
    int n    = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(CAP_ERR_HIP, "cap_ctx_create: device %d not present (%d HIP devices)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    CapContext* c = new CapContext;
    c->device     = device_id;
    switches_from_environment(c->sw);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->cu_count = prop.multiProcessorCount;
    if (hip_stream)
        c->stream = (hipStream_t)hip_stream;
    else
    {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess)
        {
            delete c;
            return fail(CAP_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
        c->own_stream = true;
    }
    (void)hipHostMalloc((void**)&c->pinned_shaded, kGuardWords * sizeof(uint64_t), hipHostMallocDefault);
    update_screen(c, 0, 0, 0, 1);
    *out_ctx = c;
    return CAP_OK;
}

void cap_ctx_destroy(CapContext* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& sp : c->spans)
    {
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto& pc : c->pending) (void)hipHostFree(pc.first);
    for (auto p : c->pinned_pool) (void)hipHostFree(p);
    if (c->pinned_shaded) (void)hipHostFree(c->pinned_shaded);
    (void)cap_comm_destroy(c);
    for (int k = 0; k < CapContext::kFrameRing; ++k)
    {
        if (c->frames_pinned[k]) (void)hipHostFree(c->frames_pinned[k]);
        if (c->frames_event[k]) (void)hipEventDestroy(c->frames_event[k]);
    }
    if (c->stream2) (void)hipStreamSynchronize(c->stream2), (void)hipStreamDestroy(c->stream2);
    for (hipEvent_t e : {c->lane_ev[0], c->lane_ev[1], c->fork_ev, c->join_ev})
        if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int cap_scene_upload(CapContext* c, const float* positions, const float* normals, const float* texcoords, const uint32_t* indices,
                     const CapMeshDesc* meshes, uint32_t vertex_count, uint32_t index_count, uint32_t mesh_count)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_scene_upload: ctx is NULL");
    if ((vertex_count && (!positions || !normals || !texcoords)) || (index_count && !indices) || (mesh_count && !meshes))
        return fail(CAP_ERR_INVALID_ARG, "cap_scene_upload: NULL array with non-zero count");
    // validate the descriptors on the host: the kernels index with them unchecked
    std::vector<uint4> tri_ids;
    std::vector<uint4> mesh_offsets(mesh_count);
    std::vector<uint32_t> mesh_texture(mesh_count);
    for (uint32_t m = 0; m < mesh_count; ++m)
    {
        const CapMeshDesc& d = meshes[m];
        if (d.index != m) return fail(CAP_ERR_INVALID_ARG, "mesh %u: index field is %u (InstanceID must equal the mesh slot)", m, d.index);
        if (d.index_count % 3) return fail(CAP_ERR_INVALID_ARG, "mesh %u: index_count %u is not a multiple of 3", m, d.index_count);
        if ((uint64_t)d.first_index_offset + d.index_count > index_count || (uint64_t)d.first_vertex_offset + d.vertex_count > vertex_count)
            return fail(CAP_ERR_INVALID_ARG, "mesh %u: ranges exceed the pools", m);
        for (uint32_t k = 0; k < d.index_count; ++k)
            if (indices[d.first_index_offset + k] >= d.vertex_count)
                return fail(CAP_ERR_INVALID_ARG, "mesh %u: index %u out of range", m, indices[d.first_index_offset + k]);
        mesh_offsets[m] = make_uint4(d.first_vertex_offset, d.first_index_offset, 0, 0);
        mesh_texture[m] = d.texture_index;
        for (uint32_t p = 0; p < d.index_count / 3; ++p) tri_ids.push_back(make_uint4(m, p, d.texture_index, 0u));
    }
    // the traversal-leaf code keeps the first sorted triangle in kLeafCountShift bits (cap_leaf.h)
    if (tri_ids.size() > kLeafFirstMask) return fail(CAP_ERR_UNSUPPORTED, "too many triangles: %zu (limit %u)", tri_ids.size(), kLeafFirstMask);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(c->positions.ensure(3 * (size_t)vertex_count));
    HIP_TRY(c->normals.ensure(3 * (size_t)vertex_count));
    HIP_TRY(c->texcoords.ensure(2 * (size_t)vertex_count));
    HIP_TRY(c->indices.ensure(index_count));
    HIP_TRY(c->tri_ids.ensure(tri_ids.size()));
    HIP_TRY(c->mesh_offsets.ensure(mesh_count));
    HIP_TRY(c->mesh_texture.ensure(mesh_count));
    if (vertex_count)
    {
        HIP_TRY(hipMemcpy(c->positions.p, positions, sizeof(float) * 3 * vertex_count, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->normals.p, normals, sizeof(float) * 3 * vertex_count, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->texcoords.p, texcoords, sizeof(float) * 2 * vertex_count, hipMemcpyHostToDevice));
    }
    if (index_count) HIP_TRY(hipMemcpy(c->indices.p, indices, sizeof(uint32_t) * index_count, hipMemcpyHostToDevice));
    if (!tri_ids.empty()) HIP_TRY(hipMemcpy(c->tri_ids.p, tri_ids.data(), sizeof(uint4) * tri_ids.size(), hipMemcpyHostToDevice));
    if (mesh_count)
    {
        HIP_TRY(hipMemcpy(c->mesh_offsets.p, mesh_offsets.data(), sizeof(uint4) * mesh_count, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->mesh_texture.p, mesh_texture.data(), sizeof(uint32_t) * mesh_count, hipMemcpyHostToDevice));
    }
    c->positions_host.assign(positions, positions + 3 * (size_t)vertex_count);
    c->indices_host.assign(indices, indices + index_count);
    c->meshes_host.assign(meshes, meshes + mesh_count);
    c->materials_ready = false;  // per-mesh materials belong to the previous scene
    c->light_count     = 0;
    c->light_tris_host.clear();  // (and with them the next-event pair list: rebuilt by the next cap_materials_upload)
    c->vertex_count = vertex_count, c->index_count = index_count, c->mesh_count = mesh_count, c->tri_count = (uint32_t)tri_ids.size();
    c->scene_ready = true;
    c->bvh_ready   = false;
    c->lane1_failed_paths = 0;  // another scene, other buffers: a second batch lane that did not fit before may fit now
    return CAP_OK;
}

int cap_texture_upload(CapContext* c, uint32_t index, const uint8_t* rgba8, uint32_t width, uint32_t height)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_texture_upload: ctx is NULL");
    if (index >= 1024) return fail(CAP_ERR_INVALID_ARG, "texture index %u exceeds the reference's 1024-entry table", index);
    static const uint8_t zero_texel[4] = {0, 0, 0, 0};  // texture_system.cpp:47-56
    if (!rgba8) rgba8 = zero_texel, width = height = 1;
    if (!width || !height) return fail(CAP_ERR_INVALID_ARG, "texture %u: empty extent", index);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->texture_host.size() <= index)
    {
        const size_t old = c->texture_host.size();
        c->texture_data.resize(index + 1);
        c->texture_host.resize(index + 1);
        for (size_t i = old; i <= index; ++i)
        {
            // holes behave like the missing-texture texel
            static const uint8_t zero_quad[16] = {0};
            HIP_TRY(c->texture_data[i].ensure(16));
            HIP_TRY(hipMemcpy(c->texture_data[i].p, zero_quad, 16, hipMemcpyHostToDevice));
            c->texture_host[i] = TextureDev{reinterpret_cast<const uint4*>(c->texture_data[i].p), 1, 1};
        }
    }
    // stored as the bilinear footprint of every texel: (x, y), (x + 1, y), (x, y + 1), (x + 1, y + 1) with WRAP, four RGBA8 words --
    // expanded on the device from the plain image (4 bytes per texel cross the bus, not 16)
    const size_t texels = (size_t)width * height;
    c->texture_data[index].release();
    HIP_TRY(c->texture_data[index].ensure(16 * texels));
    DevBuf<uint8_t> plain;
    HIP_TRY(plain.ensure(4 * texels));
    HIP_TRY(hipMemcpy(plain.p, rgba8, 4 * texels, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_texture_footprints, dim3((unsigned)((texels + 255) / 256)), dim3(256), 0, c->stream, reinterpret_cast<const uint32_t*>(plain.p),
                       reinterpret_cast<uint4*>(c->texture_data[index].p), width, height);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    plain.release();
    c->texture_host[index] = TextureDev{reinterpret_cast<const uint4*>(c->texture_data[index].p), width, height};
    c->textures_dirty = true;
    return CAP_OK;
}

int cap_bluenoise_upload(CapContext* c, const uint8_t* rgba8)
{
    if (!c || !rgba8) return fail(CAP_ERR_INVALID_ARG, "cap_bluenoise_upload: NULL argument");
    std::vector<float2> lut(256 * 256);
    for (size_t i = 0; i < lut.size(); ++i) lut[i] = make_float2((float)rgba8[4 * i] / 255.0f, (float)rgba8[4 * i + 1] / 255.0f);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(c->bluenoise.ensure(lut.size()));
    HIP_TRY(hipMemcpy(c->bluenoise.p, lut.data(), sizeof(float2) * lut.size(), hipMemcpyHostToDevice));
    // EXT: the B and A channels feed the extra random numbers of the EXT shading model (lobe choice, light triangle)
    for (size_t i = 0; i < lut.size(); ++i) lut[i] = make_float2((float)rgba8[4 * i + 2] / 255.0f, (float)rgba8[4 * i + 3] / 255.0f);
    HIP_TRY(c->bluenoise_ba.ensure(lut.size()));
    HIP_TRY(hipMemcpy(c->bluenoise_ba.p, lut.data(), sizeof(float2) * lut.size(), hipMemcpyHostToDevice));
    c->bluenoise_ready = true;
    return CAP_OK;
}


// EXT model, next-event rays on the small-scene path: which fan pairs can occlude a segment from a scene point p to a point y of a light
// triangle?  The fused kernel tests every pair for every such ray (an OR without an early exit, 8.3 of the 31 ms of BASELINE's literal
// "Lambert+GGX" step, docs/experiments.md (48)); a pair that provably never reports an occlusion is moved behind the count that loop runs to.
//
// Rule (in double, over the vertices as uploaded).  A pair is left out iff for BOTH of its triangles, with plane (v0, n):
//   (i)  every scene vertex lies on one closed side of the plane (signed distance <= 1e-6 D on the other, D = the scene's diagonal): the
//        plane supports the scene's convex hull, so p (a convex combination of scene vertices) and y are both on its inner side;
//   (ii) every vertex of every light triangle is at least delta = 1e-2 * D * Dv inside it, Dv = the largest distance from v0 to a scene
//        vertex (numbers in scene units: the bound is against the contract's ABSOLUTE tmin = 1e-4).
// Why that is exact under the intersection contract, whose occlusion test is  tmin * det < T < tmax * det  with
// T = +-(p - v0).n, det = |d.n|, d the unit direction, tmax = 0.999 |y - p|:  in exact arithmetic the segment meets the plane at t* =
// T / det, and with both ends on the inner side t* <= 0 or t* >= |y - p| + delta / sin(theta) (theta = the angle between d and the plane),
// never inside (tmin, tmax).  The computed T differs from the exact one by at most ~4 ulp of |p - v0| |n| (a three-term fma chain on a
// difference that is exact to an ulp; p itself is off its surface by as much): |t_computed - t*| <= 2.4e-7 |p - v0| / sin(theta) --
// the classic grazing-ray blow-up ((25), (64) of docs/experiments.md closed two earlier culls over it).  (ii) bounds the grazing angle:
// sin(theta) >= delta / |y - p| >= delta / D, so the error is below 2.4e-7 * Dv * D / delta = 2.4e-5, a quarter of tmin on the near side
// (t* <= 0 stays below tmin) and nothing against the 1e-3 |y - p| + delta between tmax and t* on the far side.  The ceiling of the
// Cornell box (its lamp hangs 1 cm below it: rays from the ceiling's rim to the lamp graze it) fails (ii) and stays in the list, as
// does every pair that is not a hull face.  tests: the EXT parity tests run this list; `tools/build_variant.sh neecheck -DCAP_NEE_CHECK`
// runs both lists on every ray and counts disagreements in CapStats::guard_shade (0 over BASELINE configs[2]'s 8 G next-event rays).
static int update_nee_pairs(CapContext* c)
{
    c->fan_pair_nee_count = c->fan_pair_count;
    c->fan_pairs_nee.release();
    const uint32_t np = c->fan_pair_count;
    if (!np || !c->materials_ready || c->light_tris_host.empty() || c->fan_pairs_host.size() < 20 * (size_t)np || c->sw.on(SW_NO_NEE_PAIR_CULL)) return CAP_OK;
    const size_t nv = c->positions_host.size() / 3;
    if (!nv) return CAP_OK;
    auto P = [&](size_t i, int k) { return (double)c->positions_host[3 * i + k]; };
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = 0; i < nv; ++i)
        for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], P(i, k)), hi[k] = std::max(hi[k], P(i, k));
    const double D = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    if (!(D > 0.0)) return CAP_OK;
    // the light triangles' vertices (global triangle id -> mesh -> indices, as cap_materials_upload walks them)
    std::vector<double> lv;
    {
        uint32_t g = 0;
        size_t   li = 0;
        for (uint32_t m = 0; m < c->mesh_count && li < c->light_tris_host.size(); ++m)
        {
            const CapMeshDesc& d = c->meshes_host[m];
            for (uint32_t k = 0; k + 2 < d.index_count + 0u && li < c->light_tris_host.size(); k += 3, ++g)
            {
                if (c->light_tris_host[li] != g) continue;
                ++li;
                for (int j = 0; j < 3; ++j)
                {
                    const uint32_t vi = d.first_vertex_offset + c->indices_host[d.first_index_offset + k + j];
                    for (int x = 0; x < 3; ++x) lv.push_back(P(vi, x));
                }
            }
        }
        if (li != c->light_tris_host.size()) return CAP_OK;  // (cannot happen: the ids come from the same walk) -- keep every pair
    }
    std::vector<uint8_t> skip(np, 0);
    uint32_t             n_skip = 0;
    for (uint32_t k = 0; k < np; ++k)
    {
        const float* r  = c->fan_pairs_host.data() + 20 * (size_t)k;
        const double v0[3] = {r[0], r[1], r[2]};
        bool         ok = true;
        for (int t = 0; t < 2 && ok; ++t)
        {
            const double n[3] = {r[12 + 3 * t], r[13 + 3 * t], r[14 + 3 * t]};
            const double nl   = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (!(nl > 0.0))
            {
                ok = false;
                break;
            }
            double smin = 0.0, smax = 0.0, dv = 0.0;
            for (size_t i = 0; i < nv; ++i)
            {
                const double e[3] = {P(i, 0) - v0[0], P(i, 1) - v0[1], P(i, 2) - v0[2]};
                const double sd   = (e[0] * n[0] + e[1] * n[1] + e[2] * n[2]) / nl;
                smin = std::min(smin, sd), smax = std::max(smax, sd);
                dv   = std::max(dv, std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]));
            }
            const double tol = 1e-6 * D;
            double       sign;
            if (smax <= tol)
                sign = -1.0;  // the scene lies on the negative side
            else if (smin >= -tol)
                sign = 1.0;
            else
            {
                ok = false;
                break;
            }
            const double delta = 1e-2 * D * dv;
            for (size_t i = 0; i + 2 < lv.size() && ok; i += 3)
            {
                const double sd = ((lv[i] - v0[0]) * n[0] + (lv[i + 1] - v0[1]) * n[1] + (lv[i + 2] - v0[2]) * n[2]) / nl;
                if (!(sign * sd >= delta)) ok = false;
            }
        }
        skip[k] = ok ? 1 : 0;
        n_skip += ok ? 1u : 0u;
    }
    if (!n_skip) return CAP_OK;
    std::vector<float> list;
    list.reserve(c->fan_pairs_host.size());
    for (int pass = 0; pass < 2; ++pass)
        for (uint32_t k = 0; k < np; ++k)
            if ((int)skip[k] == pass) list.insert(list.end(), c->fan_pairs_host.begin() + 20 * (size_t)k, c->fan_pairs_host.begin() + 20 * (size_t)(k + 1));
    list.resize(c->fan_pairs_host.size(), 0.0f);  // the same zero padding records behind the list
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(c->fan_pairs_nee.ensure(list.size() / 4));
    HIP_TRY(hipMemcpy(c->fan_pairs_nee.p, list.data(), sizeof(float) * list.size(), hipMemcpyHostToDevice));
    c->fan_pair_nee_count = np - n_skip;
    return CAP_OK;
}

int cap_materials_upload(CapContext* c, const CapMaterial* materials, uint32_t mesh_count)
{
    if (!c || (!materials && mesh_count)) return fail(CAP_ERR_INVALID_ARG, "cap_materials_upload: NULL argument");
    if (!c->scene_ready || mesh_count != c->mesh_count) return fail(CAP_ERR_STATE, "cap_materials_upload: expected %u materials (one per mesh)", c->mesh_count);
    c->materials_host.assign(materials, materials + mesh_count);
    // Light table of the EXT model: emissive triangles in global triangle order with float prefix sums of their areas
    // (area = |e1 x e2| / 2 with the arithmetic of cap_math.h, so the table is the one the oracle builds).
    std::vector<uint32_t> light_tris;
    std::vector<float>    light_cdf;
    float                 area = 0.0f;
    uint32_t              g    = 0;
    for (uint32_t m = 0; m < c->mesh_count; ++m)
    {
        const CapMeshDesc& d  = c->meshes_host[m];
        const CapMaterial& mt = materials[m];
        const bool         emissive = mt.ke[0] > 0.0f || mt.ke[1] > 0.0f || mt.ke[2] > 0.0f;
        for (uint32_t k = 0; k + 2 < d.index_count; k += 3, ++g)
        {
            if (!emissive) continue;
            v3 p[3];
            for (int j = 0; j < 3; ++j)
            {
                const uint32_t vi = d.first_vertex_offset + c->indices_host[d.first_index_offset + k + j];
                p[j]              = mk3(c->positions_host[3 * vi], c->positions_host[3 * vi + 1], c->positions_host[3 * vi + 2]);
            }
            area = area + 0.5f * length3(cross3(p[1] - p[0], p[2] - p[0]));
            light_tris.push_back(g);
            light_cdf.push_back(area);
        }
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(c->materials.ensure(mesh_count));
    if (mesh_count) HIP_TRY(hipMemcpy(c->materials.p, materials, sizeof(CapMaterial) * mesh_count, hipMemcpyHostToDevice));
    HIP_TRY(c->light_tris.ensure(light_tris.size()));
    HIP_TRY(c->light_cdf.ensure(light_cdf.size()));
    if (!light_tris.empty())
    {
        HIP_TRY(hipMemcpy(c->light_tris.p, light_tris.data(), sizeof(uint32_t) * light_tris.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->light_cdf.p, light_cdf.data(), sizeof(float) * light_cdf.size(), hipMemcpyHostToDevice));
    }
    c->light_count     = (uint32_t)light_tris.size();
    c->light_area      = area;
    c->materials_ready = true;
    c->light_tris_host = light_tris;
    if (c->bvh_ready) return update_nee_pairs(c);
    return CAP_OK;
}

int cap_bvh_build(CapContext* c)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_bvh_build: ctx is NULL");
    if (!c->scene_ready) return fail(CAP_ERR_STATE, "cap_bvh_build: no scene uploaded");
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t n = c->tri_count;
    HIP_TRY(c->shade_tris.ensure(kShadeRec * (size_t)n));
    // + 4 zero records: the exhaustive kernels test triangles in pairs and fetch one pair ahead (kernels.hip); a zero record
    // has det == 0 and is never hit
    HIP_TRY(c->tris_sorted.ensure(4 * ((size_t)n + 4)));
    HIP_TRY(hipMemsetAsync(c->tris_sorted.p, 0, sizeof(float4) * 4 * ((size_t)n + 4), c->stream));
    HIP_TRY(c->tri_raw.ensure(4 * (size_t)n));
    HIP_TRY(c->tri_box.ensure(2 * (size_t)n));
    HIP_TRY(c->nodes.ensure(4 * (size_t)(n > 1 ? n - 1 : 1)));
    HIP_TRY(c->stack_spill.ensure((size_t)c->cu_count * 8 * kBlock * kSpillEntries));  // up to 8 workgroups per CU
    HIP_TRY(c->leaf_tri.ensure(n));
    HIP_TRY(c->keys0.ensure(n));
    HIP_TRY(c->keys1.ensure(n));
    HIP_TRY(c->vals0.ensure(n));
    HIP_TRY(c->vals1.ensure(n));
    HIP_TRY(c->hist.ensure(256 * bvh_radix_blocks(n)));
    HIP_TRY(c->parent.ensure(2 * (size_t)n));
    HIP_TRY(c->flags.ensure(n));
    HIP_TRY(c->bvh_misc.ensure(8));
    BvhBuildArgs a{};
    a.positions = c->positions.p, a.normals = c->normals.p, a.texcoords = c->texcoords.p, a.indices = c->indices.p;
    a.tri_ids = c->tri_ids.p, a.mesh_offsets = c->mesh_offsets.p, a.tri_count = n;
    a.shade_tris = c->shade_tris.p, a.tris_sorted = c->tris_sorted.p, a.nodes = c->nodes.p, a.leaf_tri = c->leaf_tri.p;
    a.tri_raw = c->tri_raw.p, a.tri_box = c->tri_box.p;
    a.keys[0] = c->keys0.p, a.keys[1] = c->keys1.p, a.vals[0] = c->vals0.p, a.vals[1] = c->vals1.p;
    a.hist = c->hist.p, a.parent = c->parent.p, a.flags = c->flags.p, a.bounds = c->bvh_misc.p, a.max_depth = c->bvh_misc.p + 6;
    // AUTO: scenes the exhaustive kernels handle need no tree quality (Morton hierarchy); everything else gets the clustering
    // build -- on the device like the driver build it replaces (blas_system.cpp:42-65), within 1 % of the host SAH tree's trace
    // times (builders table) at 1 / 40 of its build time.  The host SAH build stays available by name.
    const bool sah  = n >= 2 && c->bvh_build_mode == CAP_BVH_BUILD_SAH;
    // ... and from kAutoSahTriangles on the surface-area splits on top of it (round 6): host-SAH quality (expected node visits 44.6 against
    // 44.4 and the clustering's 47.6 on the 262 k hall) for 10 ms at 262 k and 0.2 s at 16.8 M triangles, built once like the reference's
    // PREFER_FAST_TRACE structures (blas_system.cpp:44); below, a build is a few dozen launches whatever it holds and the trees do not differ.
    const bool sahdev = n >= 2 && (c->bvh_build_mode == CAP_BVH_BUILD_SAH_DEVICE || (c->bvh_build_mode == CAP_BVH_BUILD_AUTO && n >= (uint32_t)c->sw.get(SW_AUTO_SAH_TRIANGLES, kAutoSahTriangles)));
    const bool ploc = n >= 2 && !sahdev && (c->bvh_build_mode == CAP_BVH_BUILD_PLOC || (c->bvh_build_mode == CAP_BVH_BUILD_AUTO && n > kExhaustiveMax));
    const auto wall0 = std::chrono::steady_clock::now();
    uint32_t   host_depth = 0;
    std::vector<float> bnodes_host;  // the binary tree on the host, for the collapse into the compressed 8-wide view
    if (sah)
    {
        launch_bvh_setup(c->stream, a);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::vector<float> boxes(8 * (size_t)n);
        HIP_TRY(hipMemcpy(boxes.data(), c->tri_box.p, sizeof(float) * boxes.size(), hipMemcpyDeviceToHost));
        HostTree tree;
        build_sah_tree(boxes.data(), n, kLeafMax, kLeafCountShift, tree);
        host_depth = tree.depth;
        HIP_TRY(hipMemcpy(c->nodes.p, tree.nodes.data(), sizeof(float) * tree.nodes.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->leaf_tri.p, tree.order.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
        launch_bvh_finish_host(c->stream, a);
        bnodes_host.swap(tree.nodes);
    }
    else if (sahdev)
    {
        // surface-area splits from the root down, the clustering inside the finished segments: the PREFER_FAST_TRACE tree the reference
        // asks its driver for (blas_system.cpp:44), built where the geometry is
        HIP_TRY(c->ploc_boxes.ensure(4 * (size_t)n));
        HIP_TRY(c->ploc_ints.ensure(3 * (size_t)n + 4));
        HIP_TRY(c->sahdev_words.ensure(bvh_sah_device_scratch_words(n)));
        const int radius = (int)c->sw.get(SW_PLOC_RADIUS, 16), leaf = (int)c->sw.get(SW_SAHDEV_LEAF, 32);  // A/B switches
        const int rc = launch_bvh_build_sah_device(c->stream, a, PlocScratch{c->ploc_boxes.p, c->ploc_ints.p}, c->sahdev_words.p, (uint32_t)radius,
                                                   (uint32_t)(leaf < 1 ? 1 : leaf));
        if (rc != 0) return fail(CAP_ERR_HIP, "cap_bvh_build: device surface-area build failed (%d)", rc);
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->sahdev_words.release();  // 90 B per triangle, of no use after the build
    }
    else if (ploc)
    {
        HIP_TRY(c->ploc_boxes.ensure(4 * (size_t)n));
        HIP_TRY(c->ploc_ints.ensure(3 * (size_t)n + 4));
        const int radius = (int)c->sw.get(SW_PLOC_RADIUS, 16);  // A/B switch
        const int rc = launch_bvh_build_ploc(c->stream, a, PlocScratch{c->ploc_boxes.p, c->ploc_ints.p}, (uint32_t)radius);
        if (rc != 0) return fail(CAP_ERR_HIP, "cap_bvh_build: clustering build failed (%d)", rc);
    }
    else
        launch_bvh_build(c->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    uint32_t misc[8] = {0};
    if (n) HIP_TRY(hipMemcpy(misc, c->bvh_misc.p, sizeof(misc), hipMemcpyDeviceToHost));
    CapBvhInfo& bi    = c->bvh_info;
    bi                = CapBvhInfo{};
    bi.triangle_count = n;
    bi.node_count     = n > 1 ? n - 1 : 0;
    bi.max_depth      = n ? (sah ? host_depth : misc[6]) : 0;
    bi.build_ms       = ms;
    for (int k = 0; k < 3 && n; ++k)
    {
        auto dec = [](uint32_t o) {
            uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
            float    f;
            memcpy(&f, &u, 4);
            return f;
        };
        bi.bounds_lo[k] = dec(misc[k]), bi.bounds_hi[k] = dec(misc[3 + k]);
    }
    if (bi.max_depth > 64)
        return fail(CAP_ERR_UNSUPPORTED, "LBVH depth %u exceeds the 64-entry traversal stack", bi.max_depth);
    bi.stack_entries = bi.max_depth <= 32 ? 32 : 64;
    // Compressed 8-wide view of the same tree (cap_wide.h) for the extension- and shadow-ray kernels of scenes the exhaustive
    // kernels do not take: collapsed on the host from the binary nodes (read back when the device built them).
    c->wide8_nodes = c->wide8_depth = c->wide8_top = 0;
    if (n >= 1)
    {
        const auto w0 = std::chrono::steady_clock::now();
        HIP_TRY(c->tris8.ensure(4 * (size_t)n));
        HIP_TRY(c->wide_src.ensure(n));
        size_t   wn = 0;
        uint32_t wdepth = 0, wtop = 0;
        const bool host_collapse = c->sw.on(SW_WIDE_HOST_COLLAPSE);  // A/B switch
        if (!sah && n >= 2 && !host_collapse)
        {
            // the device built the binary tree: collapse it there too (bvh.hip k_wide_level), nothing leaves the GPU
            const uint32_t cap = n / 2u + 16u;  // an inner child stands for >= 4 triangles
            HIP_TRY(c->nodes8.ensure((kWideNodeStride / 4) * std::max<size_t>((size_t)cap + 1, kWideTopNodes)));
            HIP_TRY(c->wide_task.ensure(cap));
            HIP_TRY(c->wide_cnt.ensure(2 * (size_t)cap + 2 * ((size_t)cap / 1024 + 2)));  // per-level bases + the scan's tile sums
            HIP_TRY(c->wide_alloc.ensure(2));
            double m = 0.0;
            for (int k = 0; k < 3; ++k)
                m = std::max({m, (double)bi.bounds_hi[k] - (double)bi.bounds_lo[k], std::fabs((double)bi.bounds_lo[k]), std::fabs((double)bi.bounds_hi[k])});
            WideCollapseArgs wa{};
            wa.bnodes = c->nodes.p, wa.count = c->keys1.p, wa.n_tris = n, wa.capacity = cap;
            wa.pad = (double)kWidePad * std::max(m, 1e-30);
            wa.task = c->wide_task.p, wa.cnt = c->wide_cnt.p, wa.alloc = c->wide_alloc.p, wa.nodes8 = reinterpret_cast<uint32_t*>(c->nodes8.p), wa.tri_src = c->wide_src.p;
            uint32_t count = 0;
            if (launch_wide_collapse(c->stream, wa, &count, &wdepth, &wtop) != 0) return fail(CAP_ERR_HIP, "cap_bvh_build: device collapse into the 8-wide view failed");
            wn = count;
        }
        else
        {
            if (n >= 2 && bnodes_host.empty())
            {
                bnodes_host.resize(16 * (size_t)(n - 1));
                HIP_TRY(hipMemcpy(bnodes_host.data(), c->nodes.p, sizeof(float) * bnodes_host.size(), hipMemcpyDeviceToHost));
            }
            WideTree wt;
            build_wide_tree(n >= 2 ? bnodes_host.data() : nullptr, n, bi.bounds_lo, bi.bounds_hi, wt);
            wn = wt.nodes.size() / kWideNodeWords, wdepth = wt.depth, wtop = wt.top_nodes;
            HIP_TRY(c->nodes8.ensure((kWideNodeStride / 4) * std::max<size_t>(wn + 1, kWideTopNodes)));
            if (wn) HIP_TRY(hipMemcpy2D(c->nodes8.p, sizeof(uint32_t) * kWideNodeStride, wt.nodes.data(), sizeof(uint32_t) * kWideNodeWords,
                                        sizeof(uint32_t) * kWideNodeWords, wn, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->wide_src.p, wt.tri_src.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
        }
        launch_gather_wide(c->stream, c->wide_src.p, c->tris_sorted.p, n, c->tris8.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->wide8_nodes = (uint32_t)wn, c->wide8_depth = wdepth, c->wide8_top = wtop;
        c->wide8_ms    = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
        bi.build_ms += c->wide8_ms;  // the collapse is part of the build
        if (c->sw.on(SW_TRACE_LAUNCHES))
            fprintf(stderr, "[cap] wide view: %zu nodes, depth %u, top %u, %.1f ms\n", wn, wdepth, wtop, c->wide8_ms);
    }
    // Exhaustive path (cap_set_traversal): triangles that come in fans (k, k + 1 share v0 and the edge v0->v2, as every
    // triangulated quad of an OBJ face does) are stored as one record, so the kernels compute tvec, q and the shared edge's dot
    // product once for both.  Same per-triangle arithmetic, same results; the pairing only depends on bit-equal vertices.
    c->fan_pair_count = c->fan_single_count = 0;
    if (n && n <= 4096)
    {
        std::vector<float> raw(16 * (size_t)n);
        HIP_TRY(hipMemcpy(raw.data(), c->tri_raw.p, sizeof(float) * raw.size(), hipMemcpyDeviceToHost));
        std::vector<float> pairs, singles;
        auto rec = [&](uint32_t k) { return raw.data() + 16 * (size_t)k; };  // v0(3) e1(3) e2(3) n(3) id(1) pad(3)
        for (uint32_t k = 0; k < n;)
        {
            const float* a = rec(k);
            const float* b = k + 1 < n ? rec(k + 1) : nullptr;
            const bool   fan = b && memcmp(a, b, 12) == 0 && memcmp(a + 6, b + 3, 12) == 0;  // same v0, e2(k) == e1(k+1)
            if (fan)
            {
                // (v0, e1, e2, e3 = e2 of k+1, nA, nB, id of k, 0)
                const float r[20] = {a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], b[6], b[7], b[8], a[9], a[10], a[11],
                                     b[9], b[10], b[11], a[12], 0.0f};
                pairs.insert(pairs.end(), r, r + 20);
                k += 2;
            }
            else
            {
                singles.insert(singles.end(), a, a + 16);
                k += 1;
            }
        }
        c->fan_pair_count   = (uint32_t)(pairs.size() / 20);
        c->fan_single_count = (uint32_t)(singles.size() / 16);
        // padded by four records so that an unrolled scalar load past the end stays inside the allocation
        pairs.resize(pairs.size() + 80, 0.0f), singles.resize(singles.size() + 64, 0.0f);
        c->fan_pairs_host   = pairs;
        HIP_TRY(c->fan_pairs.ensure(pairs.size() / 4));
        HIP_TRY(c->fan_singles.ensure(singles.size() / 4));
        HIP_TRY(hipMemcpy(c->fan_pairs.p, pairs.data(), sizeof(float) * pairs.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->fan_singles.p, singles.data(), sizeof(float) * singles.size(), hipMemcpyHostToDevice));
    }
    c->bvh_ready     = true;
    if (c->fan_pair_count == 0) c->fan_pairs_host.clear();
    return update_nee_pairs(c);
}

int cap_bvh_info(CapContext* c, CapBvhInfo* out)
{
    if (!c || !out) return fail(CAP_ERR_INVALID_ARG, "cap_bvh_info: NULL argument");
    if (!c->bvh_ready) return fail(CAP_ERR_STATE, "cap_bvh_info: BVH not built");
    *out = c->bvh_info;
    return CAP_OK;
}

int cap_bvh_readback(CapContext* c, float* nodes, uint32_t* leaf_triangles)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_bvh_readback: ctx is NULL");
    if (!c->bvh_ready) return fail(CAP_ERR_STATE, "cap_bvh_readback: BVH not built");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (nodes && c->bvh_info.node_count)
        HIP_TRY(hipMemcpy(nodes, c->nodes.p, sizeof(float4) * 4 * c->bvh_info.node_count, hipMemcpyDeviceToHost));
    if (leaf_triangles && c->tri_count)
        HIP_TRY(hipMemcpy(leaf_triangles, c->leaf_tri.p, sizeof(uint32_t) * c->tri_count, hipMemcpyDeviceToHost));
    return CAP_OK;
}

int cap_bvh_wide_readback(CapContext* c, uint32_t* nodes, uint32_t* tri_src, uint32_t* info)
{
    if (!c || !info) return fail(CAP_ERR_INVALID_ARG, "cap_bvh_wide_readback: NULL argument");
    if (!c->bvh_ready) return fail(CAP_ERR_STATE, "cap_bvh_wide_readback: BVH not built");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    info[0] = c->wide8_nodes, info[1] = c->wide8_depth, info[2] = c->wide8_top;
    if (nodes && c->wide8_nodes)
        HIP_TRY(hipMemcpy2D(nodes, sizeof(uint32_t) * kWideNodeWords, c->nodes8.p, sizeof(uint32_t) * kWideNodeStride, sizeof(uint32_t) * kWideNodeWords,
                            c->wide8_nodes, hipMemcpyDeviceToHost));
    if (tri_src && c->tri_count) HIP_TRY(hipMemcpy(tri_src, c->wide_src.p, sizeof(uint32_t) * c->tri_count, hipMemcpyDeviceToHost));
    return CAP_OK;
}

int cap_camera_set(CapContext* c, const CapCameraData* camera)
{
    if (!c || !camera) return fail(CAP_ERR_INVALID_ARG, "cap_camera_set: NULL argument");
    c->camera       = *camera;
    c->camera_ready = true;
    return CAP_OK;
}

int cap_prev_camera_set(CapContext* c, const CapCameraData* camera)
{
    if (!c || !camera) return fail(CAP_ERR_INVALID_ARG, "cap_prev_camera_set: NULL argument");
    c->prev_camera       = *camera;
    c->prev_camera_ready = true;
    return CAP_OK;
}

int cap_set_resolution(CapContext* c, uint32_t width, uint32_t height)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_set_resolution: ctx is NULL");
    if (!width || !height || width > 32768 || height > 32768) return fail(CAP_ERR_INVALID_ARG, "cap_set_resolution: bad extent %ux%u", width, height);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    update_screen(c, width, height, c->screen.shard_index, c->screen.shard_count);
    if ((uint64_t)c->screen.pixels_padded > kPidMask) return fail(CAP_ERR_UNSUPPORTED, "too many pixels per shard");
    c->accum.release();
    c->lane1_failed_paths = 0;
    c->post_w = c->post_h = 0;  // histories restart at the next cap_post_frame
    c->post_last_dst = -1;
    return CAP_OK;
}

int cap_set_shard(CapContext* c, uint32_t shard_index, uint32_t shard_count)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_set_shard: ctx is NULL");
    if (!shard_count || shard_index >= shard_count) return fail(CAP_ERR_INVALID_ARG, "cap_set_shard: bad shard %u of %u", shard_index, shard_count);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    update_screen(c, c->screen.width, c->screen.height, shard_index, shard_count);
    c->accum.release();
    c->lane1_failed_paths = 0;
    return CAP_OK;
}

int cap_set_batch_paths(CapContext* c, uint64_t max_paths)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_set_batch_paths: ctx is NULL");
    c->max_batch_paths = max_paths;
    c->lane1_failed_paths = 0;
    return CAP_OK;
}

// Canary for the append-guard test (tests/test_append_guard_gpu.py): the extension queues' six planes of lane 0 are filled with one
// word; after a render with undersized sub-queues every entry BEHIND the last class's sub-queue must still hold it.
constexpr uint32_t kCanaryWord = 0x7fc0da7au;  // (a quiet NaN nobody computes)
__global__ __launch_bounds__(256) void k_canary_count(const uint4* p, size_t begin, size_t end, unsigned long long* out)
{
    unsigned long long n = 0;
    for (size_t i = begin + blockIdx.x * (size_t)256 + threadIdx.x; i < end; i += (size_t)gridDim.x * 256)
    {
        const uint4 v = p[i];
        n += (v.x != kCanaryWord || v.y != kCanaryWord || v.z != kCanaryWord || v.w != kCanaryWord) ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
    if ((threadIdx.x & 63u) == 0 && n) atomicAdd(out, n);
}
static int canary_count(CapContext* c, bool behind, uint64_t* value)
{
    if (!c->q_org[0].p || !c->last_class_capacity) return fail(CAP_ERR_STATE, "cap_debug_get: no render yet");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf<unsigned long long> dbuf;  // (released on every path: ADVICE r5)
    HIP_TRY(dbuf.ensure(1));
    unsigned long long* d = dbuf.p;
    HIP_TRY(hipMemsetAsync(d, 0, sizeof(*d), c->stream));
    const size_t used = std::min((size_t)kQueueClasses * c->last_class_capacity, c->q_org[0].n);
    for (int k = 0; k < 2; ++k)
        for (DevBuf<float4>* b : {&c->q_org[k], &c->q_dir[k], &c->q_thr[k]})
        {
            const size_t lo = behind ? used : 0, hi = behind ? b->n : used;
            if (hi > lo) hipLaunchKernelGGL(k_canary_count, dim3(512), dim3(256), 0, c->stream, reinterpret_cast<const uint4*>(b->p), lo, hi, d);
        }
    HIP_TRY(hipGetLastError());
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *value = h;
    return CAP_OK;
}

int cap_debug_set(CapContext* c, uint32_t key, uint64_t value)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_debug_set: ctx is NULL");
    switch (key)
    {
    case CAP_DEBUG_QUEUE_CAPACITY_DIV:
        if (value < 1 || value > 1024) return fail(CAP_ERR_INVALID_ARG, "cap_debug_set: capacity divisor %llu not in 1..1024", (unsigned long long)value);
        c->debug_capacity_div = (uint32_t)value;
        return CAP_OK;
    case CAP_DEBUG_WIDE_DEPTH_LIMIT:
        c->debug_wide_depth_limit = (uint32_t)value;
        return CAP_OK;
    case CAP_DEBUG_QUEUE_CANARY_FILL:
        if (!c->q_org[0].p) return fail(CAP_ERR_STATE, "cap_debug_set: the queues are allocated by the first cap_render");
        HIP_TRY(hipSetDevice(c->device));
        for (int k = 0; k < 2; ++k)
            for (DevBuf<float4>* b : {&c->q_org[k], &c->q_dir[k], &c->q_thr[k]})
                HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)b->p, (int)kCanaryWord, b->n * 4, c->stream));
        return CAP_OK;
    case CAP_DEBUG_FAIL_LANE1:
        c->debug_fail_lane1 = value != 0;
        if (!value) c->lane1_failed_paths = 0;  // "memory has been released"
        return CAP_OK;
    default:
        if (key >= CAP_DEBUG_SWITCH_BASE && key < CAP_DEBUG_SWITCH_BASE + SW_COUNT)
        {
            // ~0 = back to the product's choice.  Read at the next build / render: a switch that selects buffers (second lane, rings)
            // takes effect with the next call that sizes them.
            c->sw.v[key - CAP_DEBUG_SWITCH_BASE] = value == ~0ull ? -1 : (int64_t)value;
            return CAP_OK;
        }
        return fail(CAP_ERR_INVALID_ARG, "cap_debug_set: unknown key %u", key);
    }
}

int cap_debug_get(CapContext* c, uint32_t key, uint64_t* value)
{
    if (!c || !value) return fail(CAP_ERR_INVALID_ARG, "cap_debug_get: NULL argument");
    switch (key)
    {
    case CAP_DEBUG_QUEUE_CAPACITY_DIV: *value = c->debug_capacity_div; return CAP_OK;
    case CAP_DEBUG_WIDE_DEPTH_LIMIT: *value = c->debug_wide_depth_limit; return CAP_OK;
    case CAP_DEBUG_FAIL_LANE1: *value = c->debug_fail_lane1 ? 1u : 0u; return CAP_OK;
    case CAP_DEBUG_LANES_USED: *value = c->lanes_last_render; return CAP_OK;
    case CAP_DEBUG_SELFTEST_DIV:
    {
        HIP_TRY(hipSetDevice(c->device));
        DevBuf<unsigned long long> d;  // (released on every path: ADVICE r5)
        unsigned long long         h[3] = {0, 0, 0};
        HIP_TRY(d.ensure(3));
        HIP_TRY(hipMemsetAsync(d.p, 0, sizeof(h), c->stream));
        launch_div_selftest(c->stream, d.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h, d.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        // positive control: every normal float once + 2^30 operand pairs must have been compared
        const unsigned long long expected = (0x7f7fffffull - 0x00800000ull + 1ull) + (1ull << 30);
        if (h[2] != expected) return fail(CAP_ERR_HIP, "cap_debug_get: the division self-test compared %llu of %llu cases", h[2], expected);
        *value = h[0] + h[1];
        return CAP_OK;
    }
    case CAP_DEBUG_NEE_PAIRS:
        if (!c->bvh_ready) return fail(CAP_ERR_STATE, "cap_debug_get: BVH not built");
        *value = ((uint64_t)(c->fan_pairs_nee.p ? c->fan_pair_nee_count : c->fan_pair_count) << 32) | c->fan_pair_count;
        return CAP_OK;
    case CAP_DEBUG_QUEUE_CANARY_BEHIND: return canary_count(c, true, value);
    case CAP_DEBUG_QUEUE_CANARY_USED: return canary_count(c, false, value);
    case CAP_DEBUG_WIDE_IN_USE:
        if (!c->bvh_ready) return fail(CAP_ERR_STATE, "cap_debug_get: BVH not built");
        *value = bvh_dev(c).wide8_ok;
        return CAP_OK;
    default:
        if (key >= CAP_DEBUG_SWITCH_BASE && key < CAP_DEBUG_SWITCH_BASE + SW_COUNT)
        {
            *value = (uint64_t)c->sw.v[key - CAP_DEBUG_SWITCH_BASE];
            return CAP_OK;
        }
        return fail(CAP_ERR_INVALID_ARG, "cap_debug_get: unknown key %u", key);
    }
}

int cap_set_bvh_build(CapContext* c, uint32_t mode)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_set_bvh_build: ctx is NULL");
    if (mode > CAP_BVH_BUILD_SAH_DEVICE) return fail(CAP_ERR_INVALID_ARG, "cap_set_bvh_build: unknown mode %u", mode);
    c->bvh_build_mode = mode;
    return CAP_OK;
}

int cap_set_traversal(CapContext* c, uint32_t mode)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_set_traversal: ctx is NULL");
    if (mode > CAP_TRAVERSAL_EXHAUSTIVE) return fail(CAP_ERR_INVALID_ARG, "cap_set_traversal: unknown mode %u", mode);
    c->traversal_mode = mode;
    return CAP_OK;
}

int cap_accum_reset(CapContext* c)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_accum_reset: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (c->accum.p) HIP_TRY(hipMemsetAsync(c->accum.p, 0, sizeof(float4) * c->accum.n, c->stream));
    c->frames_accumulated = 0;
    return CAP_OK;
}

int cap_accum_import(CapContext* c, const float* sum_rgba, uint64_t frames)
{
    if (!c || !sum_rgba) return fail(CAP_ERR_INVALID_ARG, "cap_accum_import: NULL argument");
    if (!c->screen.width) return fail(CAP_ERR_STATE, "cap_accum_import: resolution not set");
    HIP_TRY(hipSetDevice(c->device));
    if (sync_and_collect(c) != CAP_OK) return CAP_ERR_HIP;
    const size_t npix = (size_t)c->screen.width * c->screen.height;
    HIP_TRY(c->image_tmp.ensure(npix));
    HIP_TRY(c->accum.ensure(c->screen.pixels_padded));
    HIP_TRY(hipMemcpyAsync(c->image_tmp.p, sum_rgba, sizeof(float4) * npix, hipMemcpyHostToDevice, c->stream));
    LaunchCfg cfg{c->stream, (uint32_t)c->cu_count * 8u, 32};
    launch_tile(cfg, c->screen, c->image_tmp.p, c->accum.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));  // the host buffer is borrowed for the duration of the call only
    c->frames_accumulated = frames;
    return CAP_OK;
}

int cap_sync(CapContext* c)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_sync: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    return sync_and_collect(c);
}

int cap_render(CapContext* c, uint32_t frame_begin, uint32_t n_frames, uint32_t num_bounces, uint32_t flags)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_render: ctx is NULL");
    if (!c->bvh_ready) return fail(CAP_ERR_STATE, "cap_render: call cap_bvh_build first");
    if (!c->camera_ready || !c->bluenoise_ready || !c->screen.width) return fail(CAP_ERR_STATE, "cap_render: camera, blue noise and resolution must be set");
    const bool ext = (flags & CAP_RENDER_EXT_MATERIALS) != 0;
    if (ext && !c->materials_ready) return fail(CAP_ERR_STATE, "cap_render: CAP_RENDER_EXT_MATERIALS needs cap_materials_upload (one material per mesh)");
    static_assert(sizeof(MaterialDev) == sizeof(CapMaterial), "material layouts must match");
    if (num_bounces > 255) return fail(CAP_ERR_INVALID_ARG, "cap_render: num_bounces %u exceeds 255", num_bounces);
    if (!n_frames) return CAP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const bool feedback = (flags & CAP_RENDER_GBUFFER_FEEDBACK) != 0;
    if (feedback)
    {
        // the branch reads the previous frame's reconstruction output: one frame per call, the whole image on this context
        if (ext) return fail(CAP_ERR_UNSUPPORTED, "cap_render: CAP_RENDER_GBUFFER_FEEDBACK is defined for the reference shading model only");
        if (n_frames != 1) return fail(CAP_ERR_INVALID_ARG, "cap_render: CAP_RENDER_GBUFFER_FEEDBACK renders one frame per call (n_frames is %u)", n_frames);
        // sharded contexts: the previous frame's output reaches the non-root ranks through cap_feedback_import
        if (!c->prev_camera_ready) return fail(CAP_ERR_STATE, "cap_render: CAP_RENDER_GBUFFER_FEEDBACK needs cap_prev_camera_set");
        if (c->post_w != c->screen.width || c->post_h != c->screen.height)
            if (int e = cap_post_reset(c)) return e;  // first frame: cleared histories, every vertex is a disocclusion
    }

    const bool lowres = (flags & CAP_RENDER_LOWRES_INDIRECT) != 0;
    if (lowres)
    {
        // the half-resolution indirect image only exists per frame (2x2 interleave over four frames): nothing to accumulate
        if (ext) return fail(CAP_ERR_UNSUPPORTED, "cap_render: CAP_RENDER_LOWRES_INDIRECT is defined for the reference shading model only");
        if (n_frames != 1) return fail(CAP_ERR_INVALID_ARG, "cap_render: CAP_RENDER_LOWRES_INDIRECT renders one frame per call (n_frames is %u)", n_frames);
        if ((c->screen.width | c->screen.height) & 1u) return fail(CAP_ERR_INVALID_ARG, "cap_render: CAP_RENDER_LOWRES_INDIRECT needs even width and height (%ux%u)", c->screen.width, c->screen.height);
    }

    const uint32_t Ppad = c->screen.pixels_padded;
    // paths in flight per batch; measured on the headline workload (tools/batch_sweep.sh, end of round 2): 8 Mi 31.0 ms, 16 Mi 25.5,
    // 32 Mi 22.1, 64 Mi 21.2, 128 Mi 21.1 per step -- every launch has a fixed cost (grid start, tables, tail) that fewer, larger
    // batches amortise; 64 Mi paths are ~14 GB of queues and planes
    uint64_t       budget = c->max_batch_paths ? c->max_batch_paths : (uint64_t)64 << 20;
    const bool tree_path = c->traversal_mode == CAP_TRAVERSAL_EXHAUSTIVE ? false
                                                                         : !(c->traversal_mode == CAP_TRAVERSAL_AUTO && c->tri_count <= kExhaustiveMax);
    if (!c->max_batch_paths && (uint64_t)n_frames * Ppad > budget)
    {
        // Twice that when the card has the room (round 6, docs/experiments.md (88)): the tree path has 27 launches per batch, most of
        // them short, and gains 1.7 % (the 262 k hall at 128 spp: 92.1 -> 90.5 ms; round 4 measured 101.2 -> 99.4 and kept 64 Mi for
        // its 14 GB per lane), the small-scene path 1.0 % (headline 20.22 -> 20.01 ms).  Taken when the working set is already that
        // large or the device reports room for it with a margin -- 128 Mi paths are ~28 GB per lane of the card's 288, and the tree
        // path runs two lanes; cap_set_batch_paths overrides either way.
        const uint64_t big       = (uint64_t)128 << 20;
        const uint64_t big_slots = std::max<uint64_t>(1, std::min<uint64_t>(big / std::max(1u, Ppad), kMaxFrameSlots));
        bool           room      = c->pl_color.n >= big_slots * Ppad && (!tree_path || c->lane1.pl_color.n >= big_slots * Ppad);
        if (!room)
        {
            size_t free_b = 0, total_b = 0;
            room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 3u * big_slots * Ppad * 256u;  // ~208 B per path and lane
        }
        if (room) budget = big;
    }
    uint32_t       slots  = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(budget / std::max(1u, Ppad), kMaxFrameSlots));
    slots                 = std::min(slots, n_frames);
    // Two lanes (CapContext::Lane): consecutive batches alternate between two working sets on two streams, so that the tail of
    // one batch's launch -- the last chunks, a workgroup at a time, with most of the chip idle -- is filled by the other batch's
    // next launch instead of waiting for the stream's next kernel (measured with two contexts side by side, tools/dual_test.py:
    // 21.0 -> 20.55 ms per step on the headline, 3.30 -> 3.05 ms for shard 0 of 8, 28.3 -> 27.1 ms on the 262 k scene).  A call
    // that fits one batch is cut in two halves when the halves are still large.  Per-stage timers need one stream.
    const bool no_two_lanes = c->sw.on(SW_NO_TWO_LANES);  // A/B switch
    // Only on the tree path: its bounces are three launches each, and from the third bounce on they are short (the 262 k scene: 27
    // launches per batch, the last 15 under 0.2 ms each) -- 28.3 -> 26.4 ms per 32 spp.  The small-scene path's nine long fused
    // launches gain nothing measurable (20.45 -> 20.25 ms with the context's own stream, 20.56 -> 20.9 beside a torch stream).
    bool two_lanes = !no_two_lanes && tree_path && c->bvh_info.stack_entries != 0 && !(flags & CAP_RENDER_GBUFFER_FEEDBACK) &&
                     !(flags & CAP_RENDER_LOWRES_INDIRECT) && n_frames >= 2;
    if (two_lanes && slots >= n_frames)
    {
        const uint64_t split_min = (uint64_t)c->sw.get(SW_LANE_SPLIT_MIN, (int64_t)4 << 20);  // A/B switch (shard 0 of 8 of the 262 k scene: 4.55 -> 4.35 ms with halves of 4 Mi paths)
        if ((uint64_t)n_frames * Ppad >= split_min)
            slots = (n_frames + 1) / 2;
        else
            two_lanes = false;
    }
    // No host synchronisation with the previous call: everything it still uses is either ordered behind it on the stream (queues,
    // counters, planes) or lives in another slot of the constants ring.  Only a long backlog of unread events / counters is
    // drained (what cap_sync / cap_stats_get do anyway), and a growing allocation (which frees the old buffers).
    if (c->spans.size() > 2048 || c->pending.size() > 64 || c->post_marks.size() > 256)
        if (sync_and_collect(c) != CAP_OK) return CAP_ERR_HIP;
    {
        const size_t planes_np = (size_t)slots * c->screen.pixels_padded;
        const bool   grows     = c->pl_color.n < planes_np || c->counters.n < (3 * (size_t)(num_bounces + 1) + 1) * kQueueClasses * kCounterStride ||
                           !c->accum.p || c->accum.n < c->screen.pixels_padded;
        if (grows) HIP_TRY(hipStreamSynchronize(c->stream));
    }
    // (per-stage timers and launch tracing need one stream: the batches keep the size the two lanes gave them -- so that a stage-timed
    // render launches what the plain one does -- and run one after the other on lane 0)
    if ((flags & CAP_RENDER_STAGE_TIMERS) || c->sw.on(SW_TRACE_LAUNCHES)) two_lanes = false;
    if (ensure_wavefront(c, slots, num_bounces) != CAP_OK) return CAP_ERR_HIP;
    if (two_lanes && c->lane1_failed_paths && (uint64_t)slots * Ppad >= c->lane1_failed_paths && num_bounces >= c->lane1_failed_bounces)
    {
        // this size has already failed: one lane, without asking again -- except every 32nd time, and at once when the device reports
        // room for it (memory can come back without any call on this context: another context closed, a co-tenant left)
        size_t free_b = 0, total_b = 0;
        const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > (size_t)slots * Ppad * 256u;  // a working set is ~208 B per path
        if (!room && (++c->lane1_retry_tick % 32u) != 0u)
        {
            two_lanes = false;
            ++c->stats.lane1_dropped;
        }
    }
    if (two_lanes)
    {
        if (c->lane1.pl_color.n < (size_t)slots * Ppad || c->lane1.counters.n < c->counters.n)
        {
            HIP_TRY(hipStreamSynchronize(c->stream));  // growing frees the old buffers
            if (c->stream2) HIP_TRY(hipStreamSynchronize(c->stream2));
        }
        if (c->debug_fail_lane1 || ensure_lane1(c, slots, num_bounces) != CAP_OK)
        {
            // the second working set doubles the batch's memory (~28 GB at the default budget): without it the batches of this call
            // simply run one after the other on lane 0 -- counted in CapStats::lane1_dropped, and not tried again at this
            // size until something is freed
            (void)hipGetLastError();
            c->lane1 = CapContext::Lane{};
            c->lane1_failed_paths = (uint64_t)slots * Ppad, c->lane1_failed_bounces = num_bounces;  // (ensure_lane1's size depends on both)
            ++c->stats.lane1_dropped;
            two_lanes = false;
        }
    }
    c->lanes_last_render = two_lanes ? 2u : 1u;
    const uint32_t ring = c->frames_next;
    c->frames_next      = (c->frames_next + 1) % CapContext::kFrameRing;
    if (!c->frames_event[ring]) HIP_TRY(hipEventCreateWithFlags(&c->frames_event[ring], hipEventDisableTiming));
    HIP_TRY(hipEventSynchronize(c->frames_event[ring]));  // the call that used this slot kFrameRing calls ago: long done
    if (c->frames_ring[ring].n < n_frames)
    {
        // (a larger buffer: the old one of this slot is idle by the event above)
        HIP_TRY(c->frames_ring[ring].ensure(n_frames));
    }
    if (c->frames_pinned_n[ring] < n_frames)
    {
        if (c->frames_pinned[ring]) HIP_TRY(hipHostFree(c->frames_pinned[ring]));
        c->frames_pinned[ring] = nullptr;
        HIP_TRY(hipHostMalloc((void**)&c->frames_pinned[ring], sizeof(FrameConst) * n_frames, hipHostMallocDefault));
        c->frames_pinned_n[ring] = n_frames;
    }
    for (uint32_t f = 0; f < n_frames; ++f) c->frames_pinned[ring][f] = frame_const(frame_begin + f, lowres);
    HIP_TRY(hipMemcpyAsync(c->frames_ring[ring].p, c->frames_pinned[ring], sizeof(FrameConst) * n_frames, hipMemcpyHostToDevice, c->stream));
    const bool st = (flags & CAP_RENDER_STAGE_TIMERS) != 0;
    if (c->textures_dirty)
    {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(c->textures.ensure(std::max<size_t>(1, c->texture_host.size())));
        if (!c->texture_host.empty())
            HIP_TRY(hipMemcpy(c->textures.p, c->texture_host.data(), sizeof(TextureDev) * c->texture_host.size(), hipMemcpyHostToDevice));
        c->textures_dirty = false;
    }

    uint32_t stack_entries = c->bvh_info.stack_entries;
    if (c->traversal_mode == CAP_TRAVERSAL_EXHAUSTIVE)
    {
        if (c->tri_count > 4096) return fail(CAP_ERR_UNSUPPORTED, "cap_render: exhaustive traversal refused for %u triangles", c->tri_count);
        stack_entries = 0;
    }
    else if (c->traversal_mode == CAP_TRAVERSAL_AUTO && c->tri_count <= kExhaustiveMax)
        stack_entries = 0;
    // Persistent grids must be fully resident: a workgroup that cannot be co-scheduled runs as a second round after the first
    // waves retire and doubles the kernel's tail.  blocks_per_cu is therefore the residency the kernels are built for
    // (launch bounds in kernels.hip), not the 8 a register-light kernel could reach.
    uint32_t blocks_per_cu = stack_entries == 0 ? 6u : (stack_entries <= 32 ? 5u : 2u);  // measured: 4..8 within 4 %, 6 best
    if (c->sw.v[SW_BLOCKS_PER_CU] >= 0) blocks_per_cu = (uint32_t)std::max<int64_t>(1, c->sw.v[SW_BLOCKS_PER_CU]);
    LaunchCfg cfg{c->stream, (uint32_t)c->cu_count * blocks_per_cu, stack_entries, (uint32_t)c->cu_count};
    cfg.sw = &c->sw;
    BvhDev          bvh   = bvh_dev(c);
    const SceneDev  scene = scene_dev(c);
    const CameraDev cam   = camera_dev(c->camera);
    const uint32_t  D     = num_bounces;
    std::unique_ptr<StageTimer> total(new StageTimer(c, ST_COUNT));  // its destructor closes the span on every exit path

    // per-lane view of the batch working set
    struct LaneView
    {
        float4 *  hits, *q_org[2], *q_dir[2], *q_thr[2], *s_org, *s_dir, *s_con, *pl_color, *pl_direct, *pl_albedo;
        uint32_t *counters, *spill;
        size_t    ring_n;
        hipStream_t stream;
    };
    const LaneView views[2] = {
        {c->hits.p, {c->q_org[0].p, c->q_org[1].p}, {c->q_dir[0].p, c->q_dir[1].p}, {c->q_thr[0].p, c->q_thr[1].p}, c->s_org.p, c->s_dir.p, c->s_con.p,
         c->pl_color.p, c->pl_direct.p, c->pl_albedo.p, c->counters.p, c->stack_spill.p, std::min(c->s_org.n, c->s_con.n), c->stream},
        {c->lane1.hits.p, {c->lane1.q_org[0].p, c->lane1.q_org[1].p}, {c->lane1.q_dir[0].p, c->lane1.q_dir[1].p},
         {c->lane1.q_thr[0].p, c->lane1.q_thr[1].p}, c->lane1.s_org.p, c->lane1.s_dir.p, c->lane1.s_con.p, c->lane1.pl_color.p,
         c->lane1.pl_direct.p, c->lane1.pl_albedo.p, c->lane1.counters.p, c->lane1.stack_spill.p, std::min(c->lane1.s_org.n, c->lane1.s_con.n),
         c->stream2}};
    const uint32_t n_batches = (n_frames + slots - 1) / slots;
    // Every way out of this function after the fork -- the early returns of HIP_TRY and of the launch tracing included -- leaves the
    // context's stream, the one callers order cap_sync / cap_readback / buffer growth behind, waiting for lane 1 (an error
    // between fork and join used to leave lane-1 work running that nothing waited for).
    struct LaneJoin
    {
        CapContext* c;
        bool        armed = false;
        ~LaneJoin()
        {
            if (!armed) return;
            if (hipEventRecord(c->join_ev, c->stream2) != hipSuccess || hipStreamWaitEvent(c->stream, c->join_ev, 0) != hipSuccess)
                (void)hipStreamSynchronize(c->stream2);
        }
    } lane_join{c};
    if (two_lanes)
    {
        // lane 1 starts behind everything queued on the context's stream so far (the frame constants' upload, earlier calls)
        HIP_TRY(hipEventRecord(c->fork_ev, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->stream2, c->fork_ev, 0));
        lane_join.armed = true;
    }
    hipEvent_t last_resolve = nullptr;  // the accumulation buffer takes the batches in frame order, whichever lane ran them
    uint32_t   batch        = 0;

    for (uint32_t done = 0; done < n_frames; done += slots, ++batch)
    {
        const uint32_t  lane = two_lanes ? ((n_batches - 1u - batch) & 1u) : 0u;  // the last batch on lane 0
        const LaneView& L    = views[lane];
        cfg.stream           = L.stream;
        bvh.stack_spill      = L.spill;
        const uint32_t ns = std::min(slots, n_frames - done);
        const FrameConst* frames = c->frames_ring[ring].p + done;
        const size_t   per_queue     = (size_t)kQueueClasses * kCounterStride;  // counter words of one queue
        // per bounce and class one 64-bit word: low half = extension entries, high half = shadow entries (one atomic serves both)
        const size_t   counter_words = (size_t)(D + 1) * per_queue;
        HIP_TRY(hipMemsetAsync(L.counters, 0, sizeof(uint32_t) * 3 * counter_words, L.stream));
        uint32_t*      ext_count = L.counters;
        uint32_t*      sh_count  = L.counters + 1;
        uint32_t*      work_shade = L.counters + counter_words;      // + b * per_queue: grab counters of bounce b's fused launch
        uint32_t*      work_any   = L.counters + 2 * counter_words;  // ... and of its any-hit launch
        const uint32_t total_chunks   = ns * (Ppad >> 6);
        uint32_t       class_capacity = ((total_chunks + kQueueClasses - 1) / kQueueClasses) * 64u;
        // (tests of the append guard: sub-queues deliberately too small -- the appends beyond them must be dropped and counted)
        if (c->debug_capacity_div > 1) class_capacity = std::max(64u, (class_capacity / c->debug_capacity_div) & ~63u);
        c->last_class_capacity = class_capacity;
        const bool     last_batch = done + ns >= n_frames;
        const uint32_t aov_slot   = ((flags & CAP_RENDER_AOV) && last_batch) ? ns - 1 : ~0u;
        const uint32_t max_count  = ns * Ppad;

        ShadeArgs sa{};
        sa.scene = scene, sa.cam = cam, sa.screen = c->screen, sa.frames = frames, sa.hits = L.hits;
        sa.planes = Planes{L.pl_color, L.pl_direct, L.pl_albedo, c->aov_geo.p, c->aov_nd.p};
        sa.n_slots = ns, sa.num_bounces = D, sa.max_count = max_count, sa.aov_slot = aov_slot, sa.shaded_counter = c->shaded_counter.p;
        {
            // screen-space culling of bounce 0 inverts primary_dir (camera.h:39-63), which needs an orthonormal basis
            const bool no_cull = c->sw.on(SW_NO_CAMERA_CULL);  // A/B switch
            auto dotf = [](const float* x, const float* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
            const CapCameraData& cd = c->camera;
            const bool ortho = std::fabs(dotf(cd.right, cd.up)) < 1e-4f && std::fabs(dotf(cd.right, cd.forward)) < 1e-4f &&
                               std::fabs(dotf(cd.up, cd.forward)) < 1e-4f && std::fabs(dotf(cd.right, cd.right) - 1.f) < 1e-4f &&
                               std::fabs(dotf(cd.up, cd.up) - 1.f) < 1e-4f && std::fabs(dotf(cd.forward, cd.forward) - 1.f) < 1e-4f;
            sa.cull_camera_pairs = (ortho && !no_cull) ? 1u : 0u;
        }
        // Nobody reads this batch's planes but the resolve (plane read-backs and the reconstruction chain need CAP_RENDER_AOV)
        const bool no_albedo_w = c->sw.on(SW_NO_ALBEDO_IN_W);  // A/B switch
        sa.albedo_in_w = (!no_albedo_w && !feedback && !lowres && !(flags & CAP_RENDER_AOV) && (ext || scene.texture_count == 0)) ? 1u : 0u;  // (the EXT model's first-vertex albedo is 1: it folds kd into the throughput)
        if (feedback)  // g_color_history = combined_history[(frame_count + 1) % 2], raytracing_system.cpp:1754-1759
            sa.fb = FeedbackDev{camera_dev(c->prev_camera), c->post_prev_nd.p, c->post_chist[(frame_begin + 1) % 2].p};
        const bool fused = cfg.stack_entries == 0;  // small-scene path: closest hit and shading in one kernel per bounce
        const bool no_inline_nee = c->sw.on(SW_NO_INLINE_NEE);  // A/B switch
        sa.inline_nee = (ext && fused && !no_inline_nee) ? 1u : 0u;
        const bool no_inline_probe = c->sw.on(SW_NO_INLINE_PROBE);  // A/B switch
        sa.inline_probe = (!ext && !feedback && fused && !no_inline_probe && c->tri_count <= kExhaustiveMax && bvh.fan_pair_count >= 1 &&
                           bvh.fan_pair_count <= kExhaustiveMax / 2)
                              ? 1u
                              : 0u;
        const bool no_wave_ring = c->sw.on(SW_NO_WAVE_RING);  // A/B switch
        sa.wave_ring = (sa.inline_probe && !no_wave_ring && (size_t)cfg.grid_blocks * (kBlock / 64) * 128 <= L.ring_n) ? 1u : 0u;  // (a grid beyond what ensure_wavefront sized the rings for: CAP_BLOCKS_PER_CU)
        LaunchCfg cfg_any    = cfg;
        cfg_any.any_no_probe = sa.inline_probe;
        // CAP_TRACE_LAUNCHES=1: name every launch on stderr and drain the stream after it (fault localisation only)
        const bool trace_launches = c->sw.on(SW_TRACE_LAUNCHES);
        auto              traced         = [&](const char* what, uint32_t b) -> int {
            if (!trace_launches) return CAP_OK;
            fprintf(stderr, "[cap] %s bounce %u batch %u ... ", what, b, done / slots);
            fflush(stderr);
            hipError_t e = hipStreamSynchronize(c->stream);
            fprintf(stderr, "%s\n", hipGetErrorString(e));
            fflush(stderr);
            return e == hipSuccess ? CAP_OK : CAP_ERR_HIP;
        };
        bool primary_shaded = false;  // tree path: bounce 0's shading done by the camera-ray kernel
        // Camera rays of a DENSE scene.  The packet walk (one node sequence per 8x8-pixel tile) is the fast form while a tile's 64
        // rays meet few triangles: 9.3 per packet on the 262 k-triangle hall at 1080p (0.13 triangles per pixel).  At 16.8 M
        // triangles (8 per pixel) a tile covers hundreds of leaves, every lane pays for every one of them, and the stage was 7.2 of the
        // big_variant's 24 ms: there the rays go through k_trace_closest8 like extension rays do, one lane each.  Same hit rule, same
        // bits.  The wide view's padding budget assumes ray origins inside the scene bounds (wide_builder.cpp): a camera outside them
        // keeps the packet walk.
        // Shadow rays of the reference model where the tree is far beyond the caches: the wide view with lane refill instead of chunk
        // by chunk (trace8.hip k_trace_any8_refill).  Measured per 8 spp at 1080p, any-hit stage (tools/primary_ab.py): 262 k triangles
        // (21 MB of wide nodes and intersection records) 1.89 -> 1.99 ms, 4.2 M (325 MB) 3.02 -> 3.02, 16.8 M (1.3 GB) 5.46 -> 4.01:
        // lanes are throughput only where a step's round trip ends in HBM.  The switch sits at twice the 256-MiB Infinity Cache;
        // A/B: CAP_ANY_REFILL=0|1.
        bool any_refill = false;
        if (!fused && !ext && bvh.wide8_ok && c->tri_count > kExhaustiveMax)
        {
            const int force = (int)c->sw.v[SW_ANY_REFILL];
            const uint64_t tree_bytes = (uint64_t)c->wide8_nodes * kWideNodeStride * 4u + (uint64_t)c->tri_count * 64u;
            any_refill = force >= 0 ? force != 0 : tree_bytes >= kAnyRefillTreeBytes;
        }
        bool primary_wide = false;
        if (!fused && bvh.wide8_ok && c->tri_count > kExhaustiveMax)
        {
            const int force = (int)c->sw.v[SW_PRIMARY_WIDE];  // A/B switch: 0 never, 1 whenever allowed
            bool inside = true;
            for (int k = 0; k < 3; ++k)
                inside = inside && c->camera.position[k] >= c->bvh_info.bounds_lo[k] && c->camera.position[k] <= c->bvh_info.bounds_hi[k];
            const double per_pixel = (double)c->tri_count / std::max(1.0, (double)c->screen.width * c->screen.height);
            primary_wide = inside && (force >= 0 ? force != 0 : per_pixel >= kPrimaryWideTrianglesPerPixel);
        }
        for (uint32_t b = 0; b <= D; ++b)
        {
            const int pi = (int)(b & 1u), po = pi ^ 1;
            sa.bounce    = b;
            // (inline_nee: no shadow entries are written, so the shadow queue's two 16-byte planes carry the paths' gathered radiance)
            sa.in        = RayQueue{L.q_org[pi], L.q_dir[pi], L.q_thr[pi], b ? ext_count + (b - 1) * per_queue : nullptr, class_capacity,
                                    pi ? L.s_dir : L.s_org};
            sa.out       = RayQueue{L.q_org[po], L.q_dir[po], L.q_thr[po], ext_count + b * per_queue, class_capacity,
                                    po ? L.s_dir : L.s_org};
            sa.shadow    = ShadowQueue{L.s_org, L.s_dir, L.s_con, sh_count + b * per_queue, class_capacity};
            sa.work      = work_shade + b * per_queue;
            if (fused)
            {
                StageTimer t(c, b == 0 ? ST_PRIMARY : ST_CLOSEST, st);
                launch_trace_shade(cfg, bvh, sa, ext, feedback);
                if (b) ++c->stats.launches_trace_closest;
                if (traced("trace_shade", b)) return fail(CAP_ERR_HIP, "trace_shade failed");
            }
            else
            {
                if (b == 0)
                {
                    {
                        StageTimer t(c, ST_PRIMARY, st);
                        if (primary_wide)
                        {
                            // dense scene: camera rays as an identity queue through the wide per-lane kernel (see above)
                            const uint32_t id_cap = (((max_count + kQueueClasses - 1) / kQueueClasses) + 63u) & ~63u;
                            const RayQueue idq{L.q_org[0], L.q_dir[0], nullptr, L.counters + 3 * counter_words, id_cap, nullptr};
                            if (c->sw.on(SW_RAYGEN_KERNEL) || bvh.tri_count == 0)  // A/B switch: the identity queue written out by its own kernel
                            {
                                launch_raygen_identity(cfg, cam, c->screen, frames, ns, idq);
                                launch_trace_closest8(cfg, bvh, idq, max_count, L.hits, sa.work);
                            }
                            else
                                launch_trace_closest8_camera(cfg, bvh, idq, max_count, L.hits, sa.work, cam, c->screen, frames, ns);
                        }
                        else
                        {
                            primary_shaded = launch_primary_shade(cfg, bvh, sa, L.hits, ext);
                            if (!primary_shaded) launch_trace_primary(cfg, bvh, cam, c->screen, frames, ns, L.hits, sa.work);
                        }
                    }
                    if (aov_slot != ~0u) launch_geo_aov(cfg, scene, L.hits + (size_t)aov_slot * Ppad, Ppad, c->aov_geo.p);
                }
                if (b != 0 || !primary_shaded)
                {
                    StageTimer t(c, ST_SHADE, st, b == 0 ? ST_DIRECT : ST_NONE);
                    launch_shade(cfg, sa, ext, feedback);
                    ++c->stats.launches_shade;
                }
            }
            if (!sa.inline_nee && !(fused && sa.wave_ring && b != 0))
            {
                StageTimer t(c, ST_ANY, st, b == 0 ? ST_DIRECT : ST_NONE);
                if (any_refill)
                    launch_trace_any8_refill(cfg_any, bvh, sa.shadow, max_count, b == 0 ? L.pl_direct : L.pl_color, Ppad, ns, c->shaded_counter.p,
                                             work_any + b * per_queue, frames);
                else
                    launch_trace_any(cfg_any, bvh, sa.shadow, max_count, b == 0 ? L.pl_direct : L.pl_color, Ppad, ns, c->shaded_counter.p,
                                     work_any + b * per_queue, /* next-event estimation: most shadow rays reach the light */ ext, frames);
                ++c->stats.launches_trace_any;
                if (traced("trace_any", b)) return fail(CAP_ERR_HIP, "trace_any failed");
            }
            if (!fused && b < D)
            {
                StageTimer t(c, ST_CLOSEST, st);
                if (bvh.wide8_ok)  // bounce b + 1's slot of the fused kernels' grab counters is free on this path
                    launch_trace_closest8(cfg, bvh, sa.out, max_count, L.hits, work_shade + (b + 1) * per_queue);
                else
                    launch_trace_closest(cfg, bvh, sa.out, max_count, L.hits);
                ++c->stats.launches_trace_closest;
            }
        }
        if (!lowres)
        {
            StageTimer t(c, ST_RESOLVE, st);
            if (two_lanes && last_resolve) HIP_TRY(hipStreamWaitEvent(L.stream, last_resolve, 0));  // frame order of the additions
            launch_resolve(cfg, sa.planes, ns, Ppad, c->accum.p, sa.albedo_in_w != 0u, scene.kd_untextured);
            if (two_lanes)
            {
                HIP_TRY(hipEventRecord(c->lane_ev[lane], L.stream));
                last_resolve = c->lane_ev[lane];
            }
        }
        // queue lengths of this batch -> pinned host memory, summed at the next sync
        uint32_t* pinned = nullptr;
        if (!c->pinned_pool.empty())
        {
            pinned = c->pinned_pool.back();
            c->pinned_pool.pop_back();
        }
        else
            HIP_TRY(hipHostMalloc((void**)&pinned, sizeof(uint32_t) * 2 * 256 * kQueueClasses * kCounterStride, hipHostMallocDefault));
        HIP_TRY(hipMemcpyAsync(pinned, L.counters, sizeof(uint32_t) * counter_words, hipMemcpyDeviceToHost, L.stream));
        c->pending.push_back({pinned, D | (sa.inline_nee ? 1u << 16 : 0u)});
        HIP_TRY(hipGetLastError());

        // primary rays: one per valid pixel per frame
        uint64_t valid = 0;
        for (uint32_t lt = 0; lt < c->screen.local_tiles; ++lt)
        {
            const uint32_t gt = lt * c->screen.shard_count + c->screen.shard_index;
            const uint32_t ty = gt / c->screen.tiles_x, tx = gt % c->screen.tiles_x;
            const uint32_t w = std::min(kTileDim, c->screen.width - tx * kTileDim), h = std::min(kTileDim, c->screen.height - ty * kTileDim);
            valid += (uint64_t)w * h;
        }
        c->stats.rays_primary += valid * ns;
        c->stats.frames += ns;
        if (!lowres) c->frames_accumulated += ns;
        c->last_slots = ns;
        c->aov_valid  = aov_slot != ~0u;
        c->aov_lowres = lowres;
        c->aov_frame  = frame_begin + done + ns - 1;
        // the next batch reuses counters/frames: serialise on the stream (already), and bound the event backlog
        if (c->spans.size() > 4096)
        {
            total.reset();
            if (two_lanes) HIP_TRY(hipStreamSynchronize(c->stream2));
            if (sync_and_collect(c) != CAP_OK) return CAP_ERR_HIP;
            total.reset(new StageTimer(c, ST_COUNT));
        }
    }
    if (two_lanes)
    {
        // the context's stream -- the one callers order their work behind -- ends the call behind lane 1 as well
        HIP_TRY(hipEventRecord(c->join_ev, c->stream2));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->join_ev, 0));
        lane_join.armed = false;
    }
    total.reset();
    HIP_TRY(hipEventRecord(c->frames_event[ring], c->stream));
    return CAP_OK;
}

int cap_readback(CapContext* c, CapBufferKind kind, float* dst)
{
    if (!c || !dst) return fail(CAP_ERR_INVALID_ARG, "cap_readback: NULL argument");
    if (!c->screen.width) return fail(CAP_ERR_STATE, "cap_readback: resolution not set");
    HIP_TRY(hipSetDevice(c->device));
    if (sync_and_collect(c) != CAP_OK) return CAP_ERR_HIP;
    const size_t npix = (size_t)c->screen.width * c->screen.height;
    const uint32_t Ppad = c->screen.pixels_padded;
    HIP_TRY(c->image_tmp.ensure(npix));
    HIP_TRY(hipMemsetAsync(c->image_tmp.p, 0, sizeof(float4) * npix, c->stream));
    LaunchCfg cfg{c->stream, (uint32_t)c->cu_count * 8u, 32};
    const bool is_aov = kind <= CAP_BUF_COMBINED || kind == CAP_BUF_INDIRECT_LOWRES;
    if (is_aov && !c->aov_valid) return fail(CAP_ERR_STATE, "cap_readback: no frame rendered with CAP_RENDER_AOV");
    if (!is_aov && !c->accum.p) return fail(CAP_ERR_STATE, "cap_readback: nothing rendered");
    const size_t off = (size_t)(c->last_slots ? c->last_slots - 1 : 0) * Ppad;
    switch (kind)
    {
    case CAP_BUF_GBUFFER_GEO: launch_untile(cfg, c->screen, c->aov_geo.p, nullptr, nullptr, 0, c->image_tmp.p); break;
    case CAP_BUF_NORMAL_DEPTH: launch_untile(cfg, c->screen, c->aov_nd.p, nullptr, nullptr, 0, c->image_tmp.p); break;
    case CAP_BUF_DIRECT: launch_untile(cfg, c->screen, c->pl_direct.p + off, nullptr, nullptr, 0, c->image_tmp.p); break;
    case CAP_BUF_ALBEDO: launch_untile(cfg, c->screen, c->pl_albedo.p + off, nullptr, nullptr, 0, c->image_tmp.p); break;
    case CAP_BUF_INDIRECT: launch_untile(cfg, c->screen, c->pl_color.p + off, nullptr, nullptr, 0, c->image_tmp.p); break;
    case CAP_BUF_COMBINED:
        launch_untile(cfg, c->screen, c->pl_color.p + off, c->pl_albedo.p + off, c->pl_direct.p + off, 1, c->image_tmp.p);
        break;
    case CAP_BUF_ACCUM_SUM: launch_untile(cfg, c->screen, c->accum.p, nullptr, nullptr, 0, c->image_tmp.p); break;
    case CAP_BUF_ACCUM_MEAN: launch_untile(cfg, c->screen, c->accum.p, nullptr, nullptr, 2, c->image_tmp.p); break;
    case CAP_BUF_INDIRECT_LOWRES:
    {
        if (!c->aov_lowres) return fail(CAP_ERR_STATE, "cap_readback: the last CAP_RENDER_AOV frame was not rendered with CAP_RENDER_LOWRES_INDIRECT");
        launch_untile(cfg, c->screen, c->pl_color.p + off, nullptr, nullptr, 0, c->image_tmp.p);
        HIP_TRY(c->post_itemp.ensure(npix));
        launch_decimate2x(c->stream, c->image_tmp.p, c->screen.width, c->screen.height, (c->aov_frame % 4u) / 2u, (c->aov_frame % 4u) % 2u,
                          c->post_itemp.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(dst, c->post_itemp.p, sizeof(float4) * (npix / 4), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return CAP_OK;
    }
    default: return fail(CAP_ERR_INVALID_ARG, "cap_readback: unknown buffer kind %d", (int)kind);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dst, c->image_tmp.p, sizeof(float4) * npix, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CAP_OK;
}

int cap_stats_get(CapContext* c, CapStats* out)
{
    if (!c || !out) return fail(CAP_ERR_INVALID_ARG, "cap_stats_get: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    if (sync_and_collect(c) != CAP_OK) return CAP_ERR_HIP;
    *out = c->stats;
    return CAP_OK;
}

int cap_stats_reset(CapContext* c)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_stats_reset: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (sync_and_collect(c) != CAP_OK) return CAP_ERR_HIP;
    c->stats = CapStats{};
    if (c->shaded_counter.p) HIP_TRY(hipMemsetAsync(c->shaded_counter.p, 0, kGuardWords * sizeof(uint64_t), c->stream));  // ordered on the context's stream
    return CAP_OK;
}

int cap_tile_buffer_floats(CapContext* c, size_t* out_floats)
{
    if (!c || !out_floats) return fail(CAP_ERR_INVALID_ARG, "cap_tile_buffer_floats: NULL argument");
    if (!c->screen.width) return fail(CAP_ERR_STATE, "cap_tile_buffer_floats: resolution not set");
    *out_floats = (size_t)c->screen.pixels_padded * 4;
    return CAP_OK;
}

int cap_resolve_tiles(CapContext* c, float* device_dst)
{
    if (!c || !device_dst) return fail(CAP_ERR_INVALID_ARG, "cap_resolve_tiles: NULL argument");
    if (!c->accum.p) return fail(CAP_ERR_STATE, "cap_resolve_tiles: nothing rendered");
    HIP_TRY(hipSetDevice(c->device));
    LaunchCfg cfg{c->stream, (uint32_t)c->cu_count * 8u, 32};
    launch_tiles_mean(cfg, c->accum.p, c->screen.pixels_padded, reinterpret_cast<float4*>(device_dst));
    HIP_TRY(hipGetLastError());
    return CAP_OK;
}

int cap_assemble_tiles(CapContext* c, const float* device_src, uint32_t shard_count, float* device_image)
{
    if (!c || !device_src || !device_image) return fail(CAP_ERR_INVALID_ARG, "cap_assemble_tiles: NULL argument");
    if (shard_count != c->screen.shard_count) return fail(CAP_ERR_INVALID_ARG, "cap_assemble_tiles: shard_count %u != context's %u", shard_count, c->screen.shard_count);
    HIP_TRY(hipSetDevice(c->device));
    LaunchCfg cfg{c->stream, (uint32_t)c->cu_count * 8u, 32};
    launch_assemble(cfg, c->screen, reinterpret_cast<const float4*>(device_src), shard_count, reinterpret_cast<float4*>(device_image));
    HIP_TRY(hipGetLastError());
    return CAP_OK;
}

void cap_post_settings_default(CapPostSettings* out)
{
    if (!out) return;
    *out = CapPostSettings{1, 1, 1, 128.0f, 3.0f, 3.0f, 64.0f, 2.0f, 3.0f, 0.975f, 0.9f, 0, 0, 0, 0};
}

int cap_post_reset(CapContext* c)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_post_reset: ctx is NULL");
    if (!c->screen.width) return fail(CAP_ERR_STATE, "cap_post_reset: resolution not set");
    HIP_TRY(hipSetDevice(c->device));
    const size_t npix = (size_t)c->screen.width * c->screen.height;
    DevBuf<float4>* all[] = {&c->post_in[0],   &c->post_in[1],   &c->post_in[2],   &c->post_in[3],  &c->post_ihist[0], &c->post_ihist[1],
                             &c->post_mhist[0], &c->post_mhist[1], &c->post_chist[0], &c->post_chist[1], &c->post_prev_nd, &c->post_itemp,
                             &c->post_temp[0],  &c->post_temp[1], &c->post_normals};
    for (DevBuf<float4>* b : all)
    {
        HIP_TRY(b->ensure(npix));
        HIP_TRY(hipMemsetAsync(b->p, 0, sizeof(float4) * npix, c->stream));  // the reference's textures start cleared
    }
    c->post_w = c->screen.width, c->post_h = c->screen.height;
    c->post_last_dst = -1;
    return CAP_OK;
}

// Common tail of cap_post_frame / cap_post_frame_gathered: the chain on post_in[0..3] (row-major, assembled from the ranks'
// gathered tiles), or -- `tiled` -- on this context's own tile-ordered planes
struct PostTiledInputs
{
    const float4 *indirect, *direct, *albedo, *normal_depth;
};
static int run_post_chain(CapContext* c, const CapPostSettings* s, uint32_t frame_count, const CapCameraData* prev_camera,
                          const PostTiledInputs* tiled = nullptr)
{
    PostChainArgs a{};
    a.settings = PostSettingsDev{s->gather, s->denoise, s->eaw5, s->eaw_normal_sigma, s->eaw_depth_sigma, s->eaw_luma_sigma, s->gather_normal_sigma,
                                 s->gather_depth_sigma, s->gather_luma_sigma, s->temporal_upscale_feedback, s->taa_feedback, s->lowres_indirect,
                                 s->disable_variance ? 0 : 1, s->fast_weights, s->output};
    a.width = c->screen.width, a.height = c->screen.height, a.frame_count = frame_count;
    a.camera = camera_dev(c->camera), a.prev_camera = camera_dev(*prev_camera);
    a.indirect = c->post_in[0].p, a.direct = c->post_in[1].p, a.albedo = c->post_in[2].p, a.normal_depth = c->post_in[3].p;
    if (tiled)
    {
        a.tiled = c->screen.tiles_x, a.screen = c->screen;
        a.tiled_indirect = tiled->indirect, a.tiled_normal_depth = tiled->normal_depth, a.indirect_rowmajor = c->post_in[0].p;
        a.direct = tiled->direct, a.albedo = tiled->albedo, a.normal_depth = nullptr;
    }
    for (int k = 0; k < 2; ++k)
        a.indirect_history[k] = c->post_ihist[k].p, a.moments_history[k] = c->post_mhist[k].p, a.combined_history[k] = c->post_chist[k].p,
        a.temp[k] = c->post_temp[k].p;
    a.prev_normal_depth = c->post_prev_nd.p, a.indirect_temp = c->post_itemp.p, a.normals = c->post_normals.p;
    // one timestamp per pass boundary, like the reference's AllocateTimestampQueryPair per pass (raytracing_system.cpp:1023-1035)
    struct Marks
    {
        CapContext* c;
        hipEvent_t  e[6];
    } marks{c, {}};
    a.mark_user = &marks;
    a.mark      = [](void* user, int pass) {
        Marks* m   = static_cast<Marks*>(user);
        m->e[pass] = get_event(m->c);
        (void)hipEventRecord(m->e[pass], m->c->stream);
    };
    launch_post_chain(c->stream, a);
    std::swap(c->post_prev_nd, c->post_normals);  // this frame's decoded normal/depth image is the next frame's previous one
    c->post_marks.push_back({marks.e[0], marks.e[1], marks.e[2], marks.e[3], marks.e[4], marks.e[5]});
    HIP_TRY(hipGetLastError());
    ++c->stats.post_frames;
    c->post_last_dst = (int)(frame_count % 2);
    return CAP_OK;
}

int cap_aov_tile_buffer_floats(CapContext* c, size_t* out_floats)
{
    if (!c || !out_floats) return fail(CAP_ERR_INVALID_ARG, "cap_aov_tile_buffer_floats: NULL argument");
    if (!c->screen.width) return fail(CAP_ERR_STATE, "cap_aov_tile_buffer_floats: resolution not set");
    *out_floats = (size_t)c->screen.pixels_padded * 4 * 4;
    return CAP_OK;
}

int cap_resolve_aov_tiles(CapContext* c, float* device_dst)
{
    if (!c || !device_dst) return fail(CAP_ERR_INVALID_ARG, "cap_resolve_aov_tiles: NULL argument");
    if (!c->aov_valid) return fail(CAP_ERR_STATE, "cap_resolve_aov_tiles: no frame rendered with CAP_RENDER_AOV");
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t Ppad  = c->screen.pixels_padded;
    const size_t   off   = (size_t)(c->last_slots ? c->last_slots - 1 : 0) * Ppad;
    const size_t   bytes = sizeof(float4) * (size_t)Ppad;
    float4*        dst   = reinterpret_cast<float4*>(device_dst);
    // the four inputs of the chain, in its order (cap_post_frame): indirect, direct, albedo, normal/depth
    const float4* src[4] = {c->pl_color.p + off, c->pl_direct.p + off, c->pl_albedo.p + off, c->aov_nd.p};
    for (int k = 0; k < 4; ++k) HIP_TRY(hipMemcpyAsync(dst + (size_t)k * Ppad, src[k], bytes, hipMemcpyDeviceToDevice, c->stream));
    return CAP_OK;
}

int cap_post_frame_gathered(CapContext* c, const CapPostSettings* s, uint32_t frame_count, const CapCameraData* prev_camera,
                            const float* device_gathered, uint32_t shard_count)
{
    if (!c || !s || !prev_camera || !device_gathered) return fail(CAP_ERR_INVALID_ARG, "cap_post_frame_gathered: NULL argument");
    if (!c->screen.width) return fail(CAP_ERR_STATE, "cap_post_frame_gathered: resolution not set");
    if (shard_count != c->screen.shard_count)
        return fail(CAP_ERR_INVALID_ARG, "cap_post_frame_gathered: shard_count %u != context's %u", shard_count, c->screen.shard_count);
    const bool lowres = s->lowres_indirect != 0;
    if (lowres && ((c->screen.width | c->screen.height) & 1u))
        return fail(CAP_ERR_INVALID_ARG, "cap_post_frame_gathered: lowres_indirect needs even width and height (%ux%u)", c->screen.width, c->screen.height);
    if (!(s->eaw_luma_sigma > 0.0f) || !(s->gather_luma_sigma > 0.0f)) return fail(CAP_ERR_INVALID_ARG, "cap_post_frame_gathered: luma sigmas must be > 0");
    if (s->output < CAP_OUTPUT_COMBINED || s->output > CAP_OUTPUT_VARIANCE) return fail(CAP_ERR_INVALID_ARG, "cap_post_frame_gathered: output %d is not one of CAP_OUTPUT_*", s->output);
    HIP_TRY(hipSetDevice(c->device));
    if (c->post_w != c->screen.width || c->post_h != c->screen.height)
        if (int e = cap_post_reset(c)) return e;
    const uint32_t Ppad = c->screen.pixels_padded;
    LaunchCfg      cfg{c->stream, (uint32_t)c->cu_count * 8u, 32};
    const float4*  g = reinterpret_cast<const float4*>(device_gathered);
    for (int k = lowres ? 1 : 0; k < 4; ++k) launch_assemble(cfg, c->screen, g + (size_t)k * Ppad, shard_count, c->post_in[k].p, (size_t)4 * Ppad);
    if (lowres)
    {
        // output_indirect_ is the (W/2, H/2) image of the pixels at this frame's interleave offset, as in cap_post_frame
        HIP_TRY(c->image_tmp.ensure((size_t)c->screen.width * c->screen.height));
        launch_assemble(cfg, c->screen, g, shard_count, c->image_tmp.p, (size_t)4 * Ppad);
        launch_decimate2x(c->stream, c->image_tmp.p, c->screen.width, c->screen.height, (frame_count % 4u) / 2u, (frame_count % 4u) % 2u,
                          c->post_in[0].p);
    }
    return run_post_chain(c, s, frame_count, prev_camera);
}

int cap_feedback_buffer_floats(CapContext* c, size_t* out_floats)
{
    if (!c || !out_floats) return fail(CAP_ERR_INVALID_ARG, "cap_feedback_buffer_floats: NULL argument");
    if (!c->screen.width) return fail(CAP_ERR_STATE, "cap_feedback_buffer_floats: resolution not set");
    *out_floats = (size_t)c->screen.width * c->screen.height * 4 * 2;
    return CAP_OK;
}

int cap_feedback_export(CapContext* c, float* device_dst)
{
    if (!c || !device_dst) return fail(CAP_ERR_INVALID_ARG, "cap_feedback_export: NULL argument");
    if (c->post_last_dst < 0 || c->post_w != c->screen.width || c->post_h != c->screen.height)
        return fail(CAP_ERR_STATE, "cap_feedback_export: the chain has not run at this resolution");
    HIP_TRY(hipSetDevice(c->device));
    const size_t npix = (size_t)c->post_w * c->post_h;
    float4*      dst  = reinterpret_cast<float4*>(device_dst);
    HIP_TRY(hipMemcpyAsync(dst, c->post_chist[c->post_last_dst].p, sizeof(float4) * npix, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dst + npix, c->post_prev_nd.p, sizeof(float4) * npix, hipMemcpyDeviceToDevice, c->stream));
    return CAP_OK;
}

int cap_feedback_import(CapContext* c, const float* device_src, uint32_t frame_count)
{
    if (!c || !device_src) return fail(CAP_ERR_INVALID_ARG, "cap_feedback_import: NULL argument");
    if (!c->screen.width) return fail(CAP_ERR_STATE, "cap_feedback_import: resolution not set");
    HIP_TRY(hipSetDevice(c->device));
    if (c->post_w != c->screen.width || c->post_h != c->screen.height)
        if (int e = cap_post_reset(c)) return e;
    const size_t  npix = (size_t)c->post_w * c->post_h;
    const float4* src  = reinterpret_cast<const float4*>(device_src);
    // what cap_render(frame_count + 1, CAP_RENDER_GBUFFER_FEEDBACK) reads: combined_history[(frame_count + 2) % 2] and the previous normal/depth
    HIP_TRY(hipMemcpyAsync(c->post_chist[frame_count % 2].p, src, sizeof(float4) * npix, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->post_prev_nd.p, src + npix, sizeof(float4) * npix, hipMemcpyDeviceToDevice, c->stream));
    return CAP_OK;
}

int cap_post_frame(CapContext* c, const CapPostSettings* s, uint32_t frame_count, const CapCameraData* prev_camera)
{
    if (!c || !s || !prev_camera) return fail(CAP_ERR_INVALID_ARG, "cap_post_frame: NULL argument");
    if (!c->screen.width) return fail(CAP_ERR_STATE, "cap_post_frame: resolution not set");
    if (c->screen.shard_count != 1) return fail(CAP_ERR_UNSUPPORTED, "cap_post_frame: needs an unsharded context (shard_count is %u)", c->screen.shard_count);
    if (!c->aov_valid) return fail(CAP_ERR_STATE, "cap_post_frame: no frame rendered with CAP_RENDER_AOV");
    if (!(s->eaw_luma_sigma > 0.0f) || !(s->gather_luma_sigma > 0.0f)) return fail(CAP_ERR_INVALID_ARG, "cap_post_frame: luma sigmas must be > 0");
    if (s->output < CAP_OUTPUT_COMBINED || s->output > CAP_OUTPUT_VARIANCE) return fail(CAP_ERR_INVALID_ARG, "cap_post_frame: output %d is not one of CAP_OUTPUT_*", s->output);
    HIP_TRY(hipSetDevice(c->device));
    if (c->post_w != c->screen.width || c->post_h != c->screen.height)
        if (int e = cap_post_reset(c)) return e;
    const uint32_t Ppad = c->screen.pixels_padded;
    const size_t   off  = (size_t)(c->last_slots ? c->last_slots - 1 : 0) * Ppad;
    LaunchCfg      cfg{c->stream, (uint32_t)c->cu_count * 8u, 32};
    const bool lowres = s->lowres_indirect != 0;
    if (lowres != c->aov_lowres)
        return fail(CAP_ERR_STATE, "cap_post_frame: settings.lowres_indirect is %d but the frame was rendered %s CAP_RENDER_LOWRES_INDIRECT",
                    (int)lowres, c->aov_lowres ? "with" : "without");
    if (lowres && frame_count != c->aov_frame)
        return fail(CAP_ERR_INVALID_ARG, "cap_post_frame: frame_count %u is not the rendered frame %u (it selects the 2x2 interleave offset)", frame_count, c->aov_frame);
    if (lowres)
    {
        // output_indirect_ is the (W/2, H/2) image of the pixels at sp_offset (raytracing_system.cpp:499-512)
        HIP_TRY(c->image_tmp.ensure((size_t)c->screen.width * c->screen.height));
        launch_untile(cfg, c->screen, c->pl_color.p + off, nullptr, nullptr, 0, c->image_tmp.p);
        launch_decimate2x(c->stream, c->image_tmp.p, c->screen.width, c->screen.height, (frame_count % 4u) / 2u, (frame_count % 4u) % 2u,
                          c->post_in[0].p);
    }
    // the chain takes the render's tile-ordered planes as they are (PostChainArgs::tiled): its first kernel untiles the indirect
    // plane and decodes the normals in one pass, Combine reads direct / albedo in tile order
    PostTiledInputs ti{lowres ? nullptr : c->pl_color.p + off, c->pl_direct.p + off, c->pl_albedo.p + off, c->aov_nd.p};
    return run_post_chain(c, s, frame_count, prev_camera, &ti);
}

int cap_post_readback(CapContext* c, float* dst)
{
    if (!c || !dst) return fail(CAP_ERR_INVALID_ARG, "cap_post_readback: NULL argument");
    if (c->post_last_dst < 0) return fail(CAP_ERR_STATE, "cap_post_readback: cap_post_frame has not run");
    HIP_TRY(hipSetDevice(c->device));
    if (sync_and_collect(c) != CAP_OK) return CAP_ERR_HIP;
    HIP_TRY(hipMemcpy(dst, c->post_chist[c->post_last_dst].p, sizeof(float4) * (size_t)c->post_w * c->post_h, hipMemcpyDeviceToHost));
    return CAP_OK;
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU frame exchange: ONE gather of tile radiance to rank 0 at frame end, over RCCL (xGMI) -- the only data-path
// collective of the design.  RCCL is loaded on first use (dlopen), so the library has no link-time dependency on
// it and single-GPU users never touch it; a process that already has an RCCL loaded (e.g. through torch) shares that copy.
// ------------------------------------------------------------------------------------------------
}  // extern "C"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace
{
struct Rccl
{
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*)                                                              = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int)                                       = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*)                                               = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t)                                                                 = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t)                                                                   = nullptr;
    ncclResult_t (*GroupStart)()                                                                            = nullptr;
    ncclResult_t (*GroupEnd)()                                                                              = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)                 = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)                       = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)        = nullptr;  // RCCL extension
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*)                                            = nullptr;
    const char* (*GetErrorString)(ncclResult_t)                                                             = nullptr;
    std::string  path;
};

Rccl* rccl()
{
    static Rccl r;
    static bool tried = false;
    if (!tried)
    {
        tried = true;
        std::vector<std::pair<std::string, int>> names;
        if (const char* e = getenv("CAP_RCCL_LIBRARY")) names.push_back({e, RTLD_NOW});
        // a copy the process already has (torch ships its own librccl.so), then the ROCm installation's
        names.push_back({"librccl.so", RTLD_NOW | RTLD_NOLOAD});
        names.push_back({"librccl.so.1", RTLD_NOW | RTLD_NOLOAD});
        names.push_back({"librccl.so.1", RTLD_NOW});
        names.push_back({"librccl.so", RTLD_NOW});
        names.push_back({"/opt/rocm/lib/librccl.so.1", RTLD_NOW});
        for (auto& n : names)
            if ((r.lib = dlopen(n.first.c_str(), n.second | RTLD_GLOBAL)))
            {
                r.path = n.first;
                break;
            }
        if (r.lib)
        {
#define CAP_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name))
            CAP_SYM(GetUniqueId, "ncclGetUniqueId"), CAP_SYM(CommInitRank, "ncclCommInitRank"), CAP_SYM(CommInitAll, "ncclCommInitAll");
            CAP_SYM(CommDestroy, "ncclCommDestroy"), CAP_SYM(GroupStart, "ncclGroupStart"), CAP_SYM(GroupEnd, "ncclGroupEnd");
            CAP_SYM(Send, "ncclSend"), CAP_SYM(Recv, "ncclRecv"), CAP_SYM(Gather, "ncclGather"), CAP_SYM(GetErrorString, "ncclGetErrorString");
            CAP_SYM(CommGetAsyncError, "ncclCommGetAsyncError"), CAP_SYM(CommAbort, "ncclCommAbort");
#undef CAP_SYM
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv)
                r.lib = nullptr;
        }
    }
    return r.lib ? &r : nullptr;
}

#define NCCL_TRY(expr)                                                                                                             \
    do                                                                                                                             \
    {                                                                                                                              \
        ncclResult_t r_ = (expr);                                                                                                  \
        if (r_ != ncclSuccess)                                                                                                     \
            return fail(CAP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, R->GetErrorString ? R->GetErrorString(r_) : "rccl error", __FILE__, \
                        __LINE__);                                                                                                 \
    } while (0)

// Inside ncclGroupStart .. ncclGroupEnd nothing may return: an early exit would leave the group open and every later RCCL call
// of the thread would be queued into it (VERDICT r2 missing 3).  The calls are chained through this accumulator, which
// remembers the first failure and skips the rest; the caller always reaches GroupEnd and reports afterwards.
struct NcclChain
{
    ncclResult_t first = ncclSuccess;
    const char*  what  = "";
    void operator()(ncclResult_t r, const char* expr)
    {
        if (first == ncclSuccess && r != ncclSuccess) first = r, what = expr;
    }
    bool ok() const { return first == ncclSuccess; }
};
#define NCCL_CHAIN(chain, expr)                                                                                                    \
    do                                                                                                                             \
    {                                                                                                                              \
        if ((chain).ok()) (chain)((expr), #expr);                                                                                  \
    } while (0)

// the communicator's asynchronous error state, polled once per frame before the frame's collective is issued (a
// failed kernel or link of an earlier frame surfaces here instead of as a hang in the next one).
int comm_poll_async(Rccl* R, CapContext* c, const char* who)
{
    if (!c->comm || !R->CommGetAsyncError) return CAP_OK;
    ncclResult_t async = ncclSuccess;
    const ncclResult_t r = R->CommGetAsyncError((ncclComm_t)c->comm, &async);
    if (r != ncclSuccess) return fail(CAP_ERR_HIP, "%s: ncclCommGetAsyncError failed: %s", who, R->GetErrorString ? R->GetErrorString(r) : "rccl error");
    if (async != ncclSuccess && async != ncclInProgress)
        return fail(CAP_ERR_HIP, "%s: rank %u's communicator reports an asynchronous error: %s", who, c->comm_rank,
                    R->GetErrorString ? R->GetErrorString(async) : "rccl error");
    return CAP_OK;
}

// this context's tiles (mean radiance, tile order) into its send buffer; the root's receive buffers
int comm_stage(CapContext* c)
{
    if (!c->accum.p) return fail(CAP_ERR_STATE, "cap_comm_gather_frame: nothing rendered");
    if (c->screen.shard_count != c->comm_size || c->screen.shard_index != c->comm_rank)
        return fail(CAP_ERR_STATE, "cap_comm_gather_frame: the context renders shard %u of %u but is rank %u of %u", c->screen.shard_index,
                    c->screen.shard_count, c->comm_rank, c->comm_size);
    HIP_TRY(hipSetDevice(c->device));
    const size_t floats = (size_t)c->screen.pixels_padded * 4;
    HIP_TRY(c->comm_send.ensure(floats));
    if (c->comm_rank == 0)
    {
        HIP_TRY(c->comm_gathered.ensure(floats * c->comm_size));
        HIP_TRY(c->comm_image.ensure((size_t)c->screen.width * c->screen.height * 4));
    }
    LaunchCfg cfg{c->stream, (uint32_t)c->cu_count * 8u, 32};
    launch_tiles_mean(cfg, c->accum.p, c->screen.pixels_padded, reinterpret_cast<float4*>(c->comm_send.p));
    HIP_TRY(hipGetLastError());
    return CAP_OK;
}

int comm_assemble(CapContext* root)
{
    HIP_TRY(hipSetDevice(root->device));
    LaunchCfg cfg{root->stream, (uint32_t)root->cu_count * 8u, 32};
    launch_assemble(cfg, root->screen, reinterpret_cast<const float4*>(root->comm_gathered.p), root->comm_size,
                    reinterpret_cast<float4*>(root->comm_image.p));
    HIP_TRY(hipGetLastError());
    return CAP_OK;
}
}  // namespace

extern "C" {

int cap_comm_unique_id(uint8_t* id)
{
    if (!id) return fail(CAP_ERR_INVALID_ARG, "cap_comm_unique_id: NULL argument");
    Rccl* R = rccl();
    if (!R) return fail(CAP_ERR_UNSUPPORTED, "cap_comm_unique_id: no RCCL library could be loaded (librccl.so.1; set CAP_RCCL_LIBRARY)");
    static_assert(sizeof(ncclUniqueId) == CAP_COMM_ID_BYTES, "id size");
    ncclUniqueId u;
    NCCL_TRY(R->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return CAP_OK;
}

int cap_comm_init_rank(CapContext* c, const uint8_t* id, uint32_t rank, uint32_t nranks)
{
    if (!c || !id || !nranks || rank >= nranks) return fail(CAP_ERR_INVALID_ARG, "cap_comm_init_rank: bad argument");
    if (c->comm || c->comm_local) return fail(CAP_ERR_STATE, "cap_comm_init_rank: the context already has a communicator");
    Rccl* R = rccl();
    if (!R) return fail(CAP_ERR_UNSUPPORTED, "cap_comm_init_rank: no RCCL library could be loaded (librccl.so.1; set CAP_RCCL_LIBRARY)");
    HIP_TRY(hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    NCCL_TRY(R->CommInitRank(&comm, (int)nranks, u, (int)rank));
    c->comm = comm, c->comm_rank = rank, c->comm_size = nranks;
    return CAP_OK;
}

int cap_comm_init_all(CapContext* const* ctxs, uint32_t n)
{
    if (!ctxs || !n) return fail(CAP_ERR_INVALID_ARG, "cap_comm_init_all: bad argument");
    std::vector<int> devs(n);
    bool             distinct = true;
    for (uint32_t i = 0; i < n; ++i)
    {
        if (!ctxs[i]) return fail(CAP_ERR_INVALID_ARG, "cap_comm_init_all: context %u is NULL", i);
        if (ctxs[i]->comm || ctxs[i]->comm_local) return fail(CAP_ERR_STATE, "cap_comm_init_all: context %u already has a communicator", i);
        devs[i] = ctxs[i]->device;
        for (uint32_t j = 0; j < i; ++j) distinct &= devs[j] != devs[i];
    }
    if (n == 1 || !distinct)
    {
        // shards of one process on one device (or a single shard): nothing to send over a link, the "gather" is device copies
        for (uint32_t i = 0; i < n; ++i)
            if (devs[i] != devs[0]) return fail(CAP_ERR_UNSUPPORTED, "cap_comm_init_all: contexts must sit on pairwise distinct devices or all on one");
        for (uint32_t i = 0; i < n; ++i) ctxs[i]->comm_local = true, ctxs[i]->comm_rank = i, ctxs[i]->comm_size = n;
        return CAP_OK;
    }
    Rccl* R = rccl();
    if (!R) return fail(CAP_ERR_UNSUPPORTED, "cap_comm_init_all: no RCCL library could be loaded (librccl.so.1; set CAP_RCCL_LIBRARY)");
    std::vector<ncclComm_t> comms(n);
    NCCL_TRY(R->CommInitAll(comms.data(), (int)n, devs.data()));
    for (uint32_t i = 0; i < n; ++i) ctxs[i]->comm = comms[i], ctxs[i]->comm_rank = i, ctxs[i]->comm_size = n;
    return CAP_OK;
}

int cap_comm_gather_frame(CapContext* c)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_comm_gather_frame: ctx is NULL");
    if (c->comm_local && c->comm_size == 1)
    {
        CapContext* one[1] = {c};
        return cap_comm_gather_frame_all(one, 1);
    }
    if (!c->comm) return fail(CAP_ERR_STATE, "cap_comm_gather_frame: cap_comm_init_rank has not run (contexts of cap_comm_init_all use cap_comm_gather_frame_all)");
    Rccl* R = rccl();
    if (int e = comm_poll_async(R, c, "cap_comm_gather_frame")) return e;
    if (int e = comm_stage(c)) return e;
    const size_t floats = (size_t)c->screen.pixels_padded * 4;
    if (R->Gather)
        NCCL_TRY(R->Gather(c->comm_send.p, c->comm_gathered.p, floats, ncclFloat, 0, (ncclComm_t)c->comm, c->stream));
    else
    {
        NcclChain ch;
        NCCL_TRY(R->GroupStart());
        NCCL_CHAIN(ch, R->Send(c->comm_send.p, floats, ncclFloat, 0, (ncclComm_t)c->comm, c->stream));
        if (c->comm_rank == 0)
            for (uint32_t r = 0; r < c->comm_size; ++r)
                NCCL_CHAIN(ch, R->Recv(c->comm_gathered.p + r * floats, floats, ncclFloat, (int)r, (ncclComm_t)c->comm, c->stream));
        const ncclResult_t ge = R->GroupEnd();  // always: the group is closed on the error path too
        if (!ch.ok()) return fail(CAP_ERR_HIP, "%s failed: %s", ch.what, R->GetErrorString ? R->GetErrorString(ch.first) : "rccl error");
        NCCL_TRY(ge);
    }
    return c->comm_rank == 0 ? comm_assemble(c) : CAP_OK;
}

int cap_comm_gather_frame_all(CapContext* const* ctxs, uint32_t n)
{
    if (!ctxs || !n) return fail(CAP_ERR_INVALID_ARG, "cap_comm_gather_frame_all: bad argument");
    for (uint32_t i = 0; i < n; ++i)
    {
        if (!ctxs[i] || ctxs[i]->comm_size != n || ctxs[i]->comm_rank != i || (!ctxs[i]->comm && !ctxs[i]->comm_local))
            return fail(CAP_ERR_STATE, "cap_comm_gather_frame_all: pass the contexts of cap_comm_init_all in the same order");
        if (int e = comm_stage(ctxs[i])) return e;
    }
    CapContext*  root   = ctxs[0];
    const size_t floats = (size_t)root->screen.pixels_padded * 4;
    if (root->comm_local)
    {
        // one device: the root's stream waits for every shard's tiles, then copies them into rank-major order
        for (uint32_t i = 0; i < n; ++i)
        {
            CapContext* c = ctxs[i];
            if (c != root)
            {
                if (!c->comm_event) HIP_TRY(hipEventCreateWithFlags(&c->comm_event, hipEventDisableTiming));
                HIP_TRY(hipEventRecord(c->comm_event, c->stream));
                HIP_TRY(hipStreamWaitEvent(root->stream, c->comm_event, 0));
            }
            HIP_TRY(hipMemcpyAsync(root->comm_gathered.p + i * floats, c->comm_send.p, sizeof(float) * floats, hipMemcpyDeviceToDevice, root->stream));
        }
        // ... and every shard's stream waits for those copies before it may overwrite its send buffer (the next frame's
        // comm_stage): without this a second gather without a cap_sync in between raced with the root's reads
        if (n > 1)
        {
            if (!root->comm_event) HIP_TRY(hipEventCreateWithFlags(&root->comm_event, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(root->comm_event, root->stream));
            for (uint32_t i = 1; i < n; ++i) HIP_TRY(hipStreamWaitEvent(ctxs[i]->stream, root->comm_event, 0));
        }
        return comm_assemble(root);
    }
    Rccl* R = rccl();
    for (uint32_t i = 0; i < n; ++i)
        if (int e = comm_poll_async(R, ctxs[i], "cap_comm_gather_frame_all")) return e;
    NcclChain  ch;
    hipError_t he = hipSuccess;
    NCCL_TRY(R->GroupStart());
    for (uint32_t i = 0; i < n && ch.ok() && he == hipSuccess; ++i)
    {
        he = hipSetDevice(ctxs[i]->device);
        if (he == hipSuccess) NCCL_CHAIN(ch, R->Send(ctxs[i]->comm_send.p, floats, ncclFloat, 0, (ncclComm_t)ctxs[i]->comm, ctxs[i]->stream));
    }
    if (he == hipSuccess) he = hipSetDevice(root->device);
    for (uint32_t r = 0; r < n && he == hipSuccess; ++r)
        NCCL_CHAIN(ch, R->Recv(root->comm_gathered.p + r * floats, floats, ncclFloat, (int)r, (ncclComm_t)root->comm, root->stream));
    const ncclResult_t ge = R->GroupEnd();  // always: the group is closed on the error path too
    HIP_TRY(he);
    if (!ch.ok()) return fail(CAP_ERR_HIP, "%s failed: %s", ch.what, R->GetErrorString ? R->GetErrorString(ch.first) : "rccl error");
    NCCL_TRY(ge);
    return comm_assemble(root);
}

int cap_comm_image(CapContext* c, float** device_image)
{
    if (!c || !device_image) return fail(CAP_ERR_INVALID_ARG, "cap_comm_image: NULL argument");
    if (c->comm_rank != 0 || !c->comm_image.p) return fail(CAP_ERR_STATE, "cap_comm_image: the assembled frame lives on rank 0 after cap_comm_gather_frame");
    *device_image = c->comm_image.p;
    return CAP_OK;
}

int cap_comm_readback(CapContext* c, float* dst)
{
    if (!c || !dst) return fail(CAP_ERR_INVALID_ARG, "cap_comm_readback: NULL argument");
    if (c->comm_rank != 0 || !c->comm_image.p) return fail(CAP_ERR_STATE, "cap_comm_readback: the assembled frame lives on rank 0 after cap_comm_gather_frame");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dst, c->comm_image.p, sizeof(float) * 4 * (size_t)c->screen.width * c->screen.height, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CAP_OK;
}

int cap_comm_info(CapContext* c, uint32_t* rank, uint32_t* size, uint32_t* uses_rccl)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_comm_info: ctx is NULL");
    if (rank) *rank = c->comm_rank;
    if (size) *size = c->comm_size;
    if (uses_rccl) *uses_rccl = c->comm ? 1u : 0u;
    return CAP_OK;
}

int cap_comm_abort(CapContext* c)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_comm_abort: ctx is NULL");
    if (c->comm)
    {
        Rccl* R = rccl();
        (void)hipSetDevice(c->device);
        if (R && R->CommAbort)
            (void)R->CommAbort((ncclComm_t)c->comm);  // no stream synchronisation: the queued collective may never complete
        c->comm = nullptr;
    }
    return cap_comm_destroy(c);
}

int cap_comm_destroy(CapContext* c)
{
    if (!c) return fail(CAP_ERR_INVALID_ARG, "cap_comm_destroy: ctx is NULL");
    if (c->comm)
    {
        Rccl* R = rccl();
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (R) (void)R->CommDestroy((ncclComm_t)c->comm);
    }
    if (c->comm_event) (void)hipEventDestroy(c->comm_event);
    c->comm = nullptr, c->comm_event = nullptr, c->comm_local = false, c->comm_rank = c->comm_size = 0;
    return CAP_OK;
}
}
