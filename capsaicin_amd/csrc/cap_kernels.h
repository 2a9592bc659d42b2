// cap_kernels.h — host-callable launchers of the gfx950 kernels (kernels.hip, bvh.hip).
#pragma once

#include "cap_device.h"

namespace cap
{
// A/B and diagnostic switches: ONE table per context (round 6; 26 getenv() calls with function-local statics before), filled from the
// environment once at cap_ctx_create and settable per context with cap_debug_set(ctx, CAP_DEBUG_SWITCH_BASE + index, value), so that a
// test or a tool flips a path without a child process.  -1 = not set: the product's own choice.  The names are the environment
// variables' (kSwitchNames, context.hip); the launchers read the table through LaunchCfg.
enum CapSwitch : uint32_t
{
    SW_NO_WIDE8 = 0,      // binary-tree kernels instead of the compressed 8-wide view
    SW_LANE1_PRIORITY,    // stream priority of the second lane
    SW_PLOC_RADIUS,       // clustering search window
    SW_SAHDEV_LEAF,       // sah_device: segments of at most this many triangles go to the clustering
    SW_WIDE_HOST_COLLAPSE,
    SW_TRACE_LAUNCHES,    // name every launch on stderr and drain the stream after it
    SW_NO_TWO_LANES,
    SW_LANE_SPLIT_MIN,
    SW_BLOCKS_PER_CU,
    SW_NO_CAMERA_CULL,
    SW_NO_ALBEDO_IN_W,
    SW_NO_INLINE_NEE,
    SW_NO_INLINE_PROBE,
    SW_NO_WAVE_RING,
    SW_ANY_REFILL,        // 0 never, 1 always: the lane-refill any-hit kernel
    SW_PRIMARY_WIDE,      // 0 never, 1 whenever allowed: camera rays through k_trace_closest8
    SW_NO_PACKET,
    SW_NO_ANY_PROBE,
    SW_ANY_PROBE,
    SW_ANY_BLOCKS,
    SW_NO_PRIMARY_FUSE,
    SW_W8_REFILL,
    SW_W8_GRID,
    SW_AUTO_SAH_TRIANGLES,  // AUTO builds with surface-area splits from this many triangles on
    SW_NO_NEE_PAIR_CULL,    // EXT model: next-event rays test every fan pair (read by the next cap_bvh_build / cap_materials_upload)
    SW_RAYGEN_KERNEL,       // dense scenes: the camera rays' identity queue written out by k_raygen_identity instead of generated in the trace kernel
    SW_COUNT
};
struct SwitchTable
{
    int64_t v[SW_COUNT];
    bool    on(CapSwitch k) const { return v[k] > 0; }              // presence flags: set and not 0
    int64_t get(CapSwitch k, int64_t dflt) const { return v[k] >= 0 ? v[k] : dflt; }
};

struct LaunchCfg
{
    hipStream_t stream;
    uint32_t    grid_blocks;    // persistent grid size for queue kernels
    uint32_t    stack_entries;  // 32 or 64 (per-lane LDS traversal stack)
    uint32_t    cu_count = 0;   // compute units (0: unknown) -- persistent kernels with a static chunk assignment clamp their grid
                                // to what is resident at once, see resident_grid() in kernels.hip
    uint32_t    any_no_probe = 0;  // launch_trace_any: the producer already probed (ShadeArgs::inline_probe): plain per-chunk kernel
    const SwitchTable* sw = nullptr;  // the context's A/B switches (null: every switch at the product's choice)
    bool    sw_on(CapSwitch k) const { return sw && sw->on(k); }
    int64_t sw_get(CapSwitch k, int64_t dflt) const { return sw ? sw->get(k, dflt) : dflt; }
};

// ---- trace ----
// Primary visibility (rt_primary_visibility.hlsl:35-49): generates camera rays for frame slots [0, n_slots) of
// the batch and writes hit records (u, v, asfloat(global triangle id | ~0u), t) at index slot * Ppad + pl.
// work: kQueueClasses zeroed chunk-grab counters (kCounterStride apart) for the persistent wide-tree variant, or NULL
void launch_trace_primary(const LaunchCfg& cfg, const BvhDev& bvh, const CameraDev& cam, const ScreenDev& screen,
                          const FrameConst* frames, uint32_t n_slots, float4* hits, uint32_t* work);
// Camera rays written as a ray queue whose entry i belongs to path i (= slot * Ppad + local pixel): q.org_tmin / q.dir_tmax hold
// n_slots * Ppad entries, q.count the 64 sub-queue counters (set by the kernel), q.class_capacity = ceil(n_slots * Ppad / 64) rounded
// up to a multiple of 64.  launch_trace_closest8 on it writes hits[i] exactly where launch_trace_primary would have.
void launch_raygen_identity(const LaunchCfg& cfg, const CameraDev& cam, const ScreenDev& screen, const FrameConst* frames, uint32_t n_slots,
                            const RayQueue& q);
// Closest hit for the extension-ray queue (rt_indirect.hlsl:173).
void launch_trace_closest(const LaunchCfg& cfg, const BvhDev& bvh, const RayQueue& q, uint32_t max_count, float4* hits);
// The same on the compressed 8-wide view of the tree (trace8.hip; needs bvh.wide8_ok).  work: kQueueClasses zeroed chunk-grab counters.
void launch_trace_closest8(const LaunchCfg& cfg, const BvhDev& bvh, const RayQueue& q, uint32_t max_count, float4* hits, uint32_t* work);
void launch_trace_closest8_camera(const LaunchCfg& cfg, const BvhDev& bvh, const RayQueue& q, uint32_t max_count, float4* hits, uint32_t* work,
                                  const CameraDev& cam, const ScreenDev& screen, const FrameConst* frames, uint32_t n_slots);
// Any hit for the shadow-ray queue (lighting.h:48-61); unoccluded rays add contrib to target[plane index].
// guard: 8 x uint64 {-, malformed path ids seen by shade, by trace_any, last offender, appends beyond a class's capacity, -, -, -}
// work: kQueueClasses zeroed chunk-grab counters for this launch (exhaustive path; may be NULL for the LBVH kernels)
void launch_trace_any(const LaunchCfg& cfg, const BvhDev& bvh, const ShadowQueue& q, uint32_t max_count, float4* target,
                      uint32_t pixels_padded, uint32_t n_slots, uint64_t* guard, uint32_t* work, bool mostly_unoccluded,
                      const FrameConst* frames);

// The same for the reference model's shadow rays on the wide view with lane refill (trace8.hip): dense scenes, where a traversal step's
// round trip ends in HBM.  Needs bvh.wide8_ok and the zeroed grab counters `work`.
void launch_trace_any8_refill(const LaunchCfg& cfg, const BvhDev& bvh, const ShadowQueue& q, uint32_t max_count, float4* target, uint32_t pixels_padded,
                              uint32_t n_slots, uint64_t* guard, uint32_t* work, const FrameConst* frames);

// ---- shade ----
struct ShadeArgs
{
    SceneDev          scene;
    CameraDev         cam;
    ScreenDev         screen;
    const FrameConst* frames;
    const float4*     hits;
    RayQueue          in;       // unused for the first bounce (identity queue)
    RayQueue          out;
    ShadowQueue       shadow;
    Planes            planes;
    uint32_t          n_slots;     // frame slots in this batch
    uint32_t          bounce;
    uint32_t          num_bounces;
    uint32_t          max_count;   // upper bound of the input queue length
    uint32_t          aov_slot;    // frame slot whose AOVs are kept, or ~0u
    uint64_t*         shaded_counter;
    uint32_t*         work;        // fused kernels: kQueueClasses chunk-grab counters of this launch (zeroed), kCounterStride apart
    FeedbackDev       fb;          // read only by the feedback variants
    // untextured scene, reference shading, accumulate-only render (nobody but the resolve reads the planes): the first vertex's albedo
    // is one of four constants, so the albedo plane is not used and direct.w carries a code instead of 1 (kernels.hip shade_vertex)
    uint32_t          albedo_in_w;
    // EXT model on the small-scene path: the next-event shadow ray is tested inside the fused kernel (same exhaustive loop as the
    // any-hit kernel's) and the path carries its gathered radiance in the extension queue (RayQueue::acc); the colour plane is
    // written once, when the path ends, the direct plane once at bounce 0 -- no shadow queue, no any-hit launch, no scattered
    // read-modify-write.  The sums are the same additions in the same (bounce) order.
    uint32_t          inline_nee;
    // Reference model on the small-scene path: the fused kernel tests every shadow ray it generates against ONE fan pair -- the one
    // farthest along the batch's first light direction, which occludes most of them -- and only queues the survivors for the any-hit
    // kernel (which then tests every pair, without a probe of its own).  Occlusion is an OR over the pairs: same result.
    // The shadow rays the probe answered are rays of the statistics, not entries: the launch adds their number to word 2 of its
    // classes' counter lines (flush_stats).
    uint32_t          inline_probe;
    // ... and from bounce 1 on the survivors are traced by the wave that found them (k_trace_shade's per-wave ring): the only
    // any-hit launch left on the small-scene path is bounce 0's
    uint32_t          wave_ring;
    uint32_t          cull_camera_pairs;  // bounce 0 of the small-scene path: the camera basis is orthonormal, so a tile may skip the pairs off its screen area
};
// feedback: vertices of bounce >= 1 that the previous frame saw take its shaded colour and end the path (rt_indirect.hlsl:116-145;
// reference shading model only)
void launch_shade(const LaunchCfg& cfg, const ShadeArgs& args, bool ext, bool feedback = false);
// tree path, bounce 0: camera-ray packet walk + shading in one kernel (args.work = bounce 0's grab counters); false if the
// configuration does not take the packet walk -- then launch_trace_primary + launch_shade do the same in two kernels
bool launch_primary_shade(const LaunchCfg& cfg, const BvhDev& bvh, const ShadeArgs& args, float4* hits, bool ext);
// small-scene path: exhaustive closest hit fused with the shading of the vertex found (bounce 0 generates the camera rays)
void launch_trace_shade(const LaunchCfg& cfg, const BvhDev& bvh, const ShadeArgs& args, bool ext, bool feedback = false);

// ---- accumulate / exchange ----
// accum[pl] += sum over slots (in slot order) of color*albedo + direct; .w counts frames.
void launch_resolve(const LaunchCfg& cfg, const Planes& planes, uint32_t n_slots, uint32_t pixels_padded, float4* accum,
                    bool albedo_in_w = false, float kd_untextured = 0.0f);
// plane_kind: 0 copy, 1 combined (color*albedo+direct from the three planes at slot offset), 2 mean (xyz / w)
void launch_untile(const LaunchCfg& cfg, const ScreenDev& screen, const float4* src, const float4* albedo, const float4* direct,
                   int plane_kind, float4* image);
// the inverse of launch_untile(.., 0, ..): a row-major image into this shard's tile order (cap_accum_import)
void launch_tile(const LaunchCfg& cfg, const ScreenDev& screen, const float4* image, float4* dst);
// s0 may be null (its image is produced elsewhere)
void launch_untile4(const LaunchCfg& cfg, const ScreenDev& screen, const float4* s0, const float4* s1, const float4* s2, const float4* s3,
                    float4* d0, float4* d1, float4* d2, float4* d3);
void launch_tiles_mean(const LaunchCfg& cfg, const float4* accum, uint32_t pixels_padded, float4* dst);
// shard_stride: float4 elements between two shards' buffers (0: pixels_padded, i.e. back to back)
void launch_assemble(const LaunchCfg& cfg, const ScreenDev& screen, const float4* gathered, uint32_t shard_count, float4* image,
                     size_t shard_stride = 0);
void launch_geo_aov(const LaunchCfg& cfg, const SceneDev& scene, const float4* hits_slot, uint32_t pixels_padded, float4* aov_geo);

// ---- LBVH build (bvh.hip) ----
struct BvhBuildArgs
{
    // inputs: GeometryStorage layout on the device
    const float*    positions;
    const float*    normals;
    const float*    texcoords;
    const uint32_t* indices;
    const uint4*    tri_ids;        // (instance, primitive, texture index, -) per global triangle
    const uint4*    mesh_offsets;   // per mesh: (first_vertex_offset, first_index_offset, -, -)
    uint32_t        tri_count;
    // outputs
    float4*         shade_tris;     // kShadeRec per triangle, global order
    float4*         tris_sorted;    // 4 per triangle, leaf order
    float4*         nodes;          // 4 per internal node
    uint32_t*       leaf_tri;       // global triangle id per leaf
    // scratch
    float4*         tri_raw;        // 4 per triangle, global order
    float4*         tri_box;        // 2 per triangle (lo, hi), global order
    uint32_t*       keys[2];
    uint32_t*       vals[2];
    uint32_t*       hist;           // radix histogram scratch: 256 * blocks
    uint32_t*       parent;         // [2*tri_count] parents of internal nodes then leaves, (parent << 1) | slot
    uint32_t*       flags;          // [tri_count] refit arrival counters
    uint32_t*       bounds;         // 6 orderable-uint encoded floats
    uint32_t*       max_depth;      // 1
};
size_t bvh_radix_blocks(uint32_t n);
void   launch_bvh_build(hipStream_t stream, const BvhBuildArgs& a);
int    launch_bvh_sort(hipStream_t stream, const BvhBuildArgs& a);  // setup + Morton order only: a.keys[r] / a.vals[r] sorted, returns r
// Agglomerative build with a surface-area distance over the Morton order (ploc.hip); same outputs as launch_bvh_build,
// incl. the subtree counts in a.keys[1].  boxes: 4 * tri_count float4; ints: 3 * tri_count + 4 words.  Returns 0 on success.
struct PlocScratch
{
    float4*   boxes;
    uint32_t* ints;
};
int    launch_bvh_build_ploc(hipStream_t stream, const BvhBuildArgs& a, const PlocScratch& s, uint32_t radius);
// Binned surface-area splits from the root down to segments of <= `leaf` triangles, the clustering inside those (ploc.hip,
// "sah_device"); same outputs and the same PlocScratch, plus bvh_sah_device_scratch_words(n) words of its own.
size_t bvh_sah_device_scratch_words(uint32_t n);
int    launch_bvh_build_sah_device(hipStream_t stream, const BvhBuildArgs& a, const PlocScratch& s, uint32_t* scratch, uint32_t radius, uint32_t leaf);
// Host-built tree (sah_builder.cpp): setup = triangle records, boxes and scene bounds only; finish = after `nodes` and
// `leaf_tri` have been uploaded: intersection records into leaf order + the wide view.
void launch_bvh_setup(hipStream_t stream, const BvhBuildArgs& a);
void launch_bvh_finish_host(hipStream_t stream, const BvhBuildArgs& a);

// Compressed 8-wide view (wide_builder.cpp builds the nodes on the host): intersection records into its leaf order.
void     launch_gather_wide(hipStream_t stream, const uint32_t* tri_src, const float4* tris_sorted, uint32_t n, float4* tris8);
// Device-side collapse of the device-built binary tree (needs BvhBuildArgs::keys[1] = the subtree counts launch_bvh_build leaves
// there).  task: capacity words of scratch; cnt: 2 * capacity; alloc: 2 words; nodes8: capacity * kWideNodeStride words; tri_src: n_tris words.
struct WideCollapseArgs
{
    const float4*   bnodes;
    const uint32_t* count;
    uint32_t        n_tris, capacity;
    double          pad;
    uint32_t *      task, *alloc, *nodes8, *tri_src;
    uint32_t*       cnt;  // 2 * capacity + 2 * (capacity / 1024 + 2) words: per node of the level being written, (inner children, triangles) -> their bases; the scan's tile sums behind
    uint32_t        begin, end;
};
int      launch_wide_collapse(hipStream_t stream, WideCollapseArgs a, uint32_t* node_count, uint32_t* depth, uint32_t* top_nodes);
uint32_t wide8_stack_pairs();  // (g_base, g_mask) entries a lane of the wide kernels can hold: the tree's depth - 1 must fit

// ---- reconstruction chain (post.hip): Gather -> Accumulate -> BlurDisocclusion -> Blur -> Combine -> TAA ----
struct PostSettingsDev  // SettingsComponent subset, gui_system.h:20-37
{
    int   gather, denoise, eaw5;
    float eaw_normal_sigma, eaw_depth_sigma, eaw_luma_sigma;
    float gather_normal_sigma, gather_depth_sigma, gather_luma_sigma;
    float temporal_upscale_feedback, taa_feedback;
    int   lowres_indirect;  // UPSCALE2X: `indirect` is the (W/2, H/2) image of this frame's interleave offset
    int   use_variance;     // USE_VARIANCE of eaw_blur.hlsl
    int   fast_weights;     // hardware exp / log / rcp in the edge-stopping weights (toleranced mode)
    int   output;           // SettingsComponent::output = CombineIllumination's `type` (combine_illumination.hlsl:26-40): 0..3
};
struct PostChainArgs
{
    PostSettingsDev settings;
    uint32_t        width, height, frame_count;
    CameraDev       camera, prev_camera;
    // this frame's ray-pass outputs, row-major W*H ...
    const float4 *indirect, *direct, *albedo, *normal_depth;
    // ... or, on an unsharded context (cap_post_frame), straight from the render's tile-ordered planes: `tiled` = the screen's tile
    // columns (0 = row-major inputs as above).  Then `tiled_indirect` / `tiled_normal_depth` are untiled by the chain's first
    // kernel -- which decodes the normals on the way, so `normal_depth` is not needed row-major at all -- and Combine reads
    // `direct` / `albedo` in tile order: three image round trips less per frame.
    uint32_t      tiled;
    const float4 *tiled_indirect, *tiled_normal_depth;
    float4*       indirect_rowmajor;  // where the untiled indirect plane goes (the chain's `indirect` when tiled)
    ScreenDev     screen;
    // persistent state (raytracing_system.cpp:262-317)
    float4 *indirect_history[2], *moments_history[2], *combined_history[2], *prev_normal_depth;
    // scratch
    float4 *indirect_temp, *temp[2], *normals;  // normals: decoded (n.xyz, depth) of this frame
    // called on the host before pass p's launches, p = 0..4: Spatial gather, Temporal upscale, EAW, Combine illumination, TAA,
    // and with p = 5 after the last launch (per-pass timestamps like the reference's AllocateTimestampQueryPair); may be null
    void (*mark)(void* user, int pass);
    void* mark_user;
};
// The frame's output is combined_history[frame_count % 2] (raytracing_system.cpp:320-324).
void launch_post_chain(hipStream_t stream, const PostChainArgs& a);
// post.hip's unscaled IEEE division against the compiler's, on the device: out[0] mismatches of log2 over every normal float,
// out[1] over 2^30 operand pairs of the range it is used on (both must be 0; cap_debug_get(CAP_DEBUG_SELFTEST_DIV))
void launch_div_selftest(hipStream_t stream, unsigned long long* out_device);
// out[(y, x)] = full[(2y + oy, 2x + ox)]: the half-resolution indirect image of LOWRES_INDIRECT (rt_indirect.hlsl:53-59, :176)
void launch_decimate2x(hipStream_t stream, const float4* full, uint32_t width, uint32_t height, uint32_t ox, uint32_t oy, float4* out);
}  // namespace cap
