/*
 * capsaicin_hip.h — C ABI of the MI355X (gfx950) wavefront path tracer: the device-side drop-in for the
 * reference's RaytracingSystem + BLAS/TLAS systems.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Each entry point names the reference interface it replaces (paths relative to /root/reference/src/core).
 * All functions return CAP_OK (0) or a CapStatus error; cap_last_error() gives the thread-local message.
 * No exception crosses this boundary.  A context is single-threaded (the reference runs every system on the
 * UI thread, capsaicin.cpp:38-45, main.cpp:17-19); use one context per GPU.
 * Host pointers are borrowed for the duration of the call only.  Pointers documented as "device" must be
 * device-accessible on the context's GPU.
 */
#ifndef CAPSAICIN_HIP_H
#define CAPSAICIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct CapContext CapContext;

typedef enum CapStatus
{
    CAP_OK                = 0,
    CAP_ERR_INVALID_ARG   = 1,
    CAP_ERR_HIP           = 2, /* a HIP runtime call failed (no GPU, out of memory, ...) */
    CAP_ERR_STATE         = 3, /* call order violated (e.g. render before bvh_build) */
    CAP_ERR_UNSUPPORTED   = 4,
    CAP_ERR_IO            = 5
} CapStatus;

/* CameraData, src/systems/camera_system.h:16-31 == shaders/data_payload.h:7-18 (72 bytes). */
typedef struct CapCameraData
{
    float position[3];
    float focal_length;
    float right[3];
    float znear;
    float forward[3];
    float focus_distance;
    float up[3];
    float aperture;
    float sensor_size[2];
} CapCameraData;

/* MeshComponent, src/systems/asset_load_system.h:29-39 == shaders/data_payload.h:20-30 (32 bytes). */
typedef struct CapMeshDesc
{
    uint32_t vertex_count;
    uint32_t first_vertex_offset;
    uint32_t index_count;
    uint32_t first_index_offset;
    uint32_t index;
    uint32_t texture_index; /* MeshComponent::material_index; ~0u = untextured */
    uint32_t padding[2];
} CapMeshDesc;

/* EXT (no reference counterpart): per-mesh material for CAP_RENDER_EXT_MATERIALS. */
typedef struct CapMaterial
{
    float kd[3];
    float roughness;
    float ks[3];
    float pad0;
    float ke[3];
    float pad1;
} CapMaterial;

/* cap_render flags */
enum
{
    CAP_RENDER_AOV           = 1u << 0, /* keep the per-frame planes of the LAST frame for cap_readback */
    CAP_RENDER_EXT_MATERIALS = 1u << 1, /* EXT shading model (materials uploaded with cap_materials_upload) */
    CAP_RENDER_STAGE_TIMERS  = 1u << 2, /* bracket every kernel with hipEvents (fills CapStats::ms_<stage>) */
    /* RaytracingOptions::gbuffer_feedback (raytracing_system.h:26, rt_indirect.hlsl:116-145): a path vertex of bounce >= 1
     * that the previous frame saw takes that frame's cap_post_frame output and ends the path.  One frame per call, reference
     * shading model (sharded contexts: cap_feedback_export / _import carry the previous output to every rank); needs
     * cap_prev_camera_set and cap_post_frame after every frame. */
    CAP_RENDER_GBUFFER_FEEDBACK = 1u << 3,
    /* RaytracingOptions::lowres_indirect (raytracing_system.h:24; LOWRES_INDIRECT, rt_indirect.hlsl:53-59): only the pixel at
     * sp_offset = ((frame % 4) / 2, (frame % 4) % 2) of every 2x2 block gets an indirect sample.  One frame per call, even width
     * and height, reference shading model; the frame is not added to the accumulation buffer.  Read the
     * (W/2, H/2) image with CAP_BUF_INDIRECT_LOWRES or hand it to cap_post_frame (settings.lowres_indirect = 1). */
    CAP_RENDER_LOWRES_INDIRECT = 1u << 4
};

/* cap_readback kinds: the reference's RaytracingSystem outputs (raytracing_system.h, cpp:466-575). */
typedef enum CapBufferKind
{
    CAP_BUF_GBUFFER_GEO  = 0, /* rt_primary_visibility.hlsl:46  (u, v, asfloat(instance), asfloat(prim)) */
    CAP_BUF_DIRECT       = 1, /* rt_direct_lighting.hlsl:78      output_direct_                           */
    CAP_BUF_ALBEDO       = 2, /* rt_direct_lighting.hlsl:79      gbuffer_albedo_                          */
    CAP_BUF_NORMAL_DEPTH = 3, /* rt_direct_lighting.hlsl:80      gbuffer_normal_depth_                    */
    CAP_BUF_INDIRECT     = 4, /* rt_indirect.hlsl:176            output_indirect_                         */
    CAP_BUF_COMBINED     = 5, /* combine_illumination.hlsl:29    indirect*albedo + direct (xyz; w = 1)    */
    CAP_BUF_ACCUM_SUM    = 6, /* running fp32 sum of COMBINED over all frames since cap_accum_reset (w = frames) */
    CAP_BUF_ACCUM_MEAN   = 7, /* ACCUM_SUM / frames                                                       */
    CAP_BUF_INDIRECT_LOWRES = 8 /* output_indirect_ of a CAP_RENDER_LOWRES_INDIRECT frame: (width/2)*(height/2)*4 floats */
} CapBufferKind;

/* Per-stage names follow the reference's timestamp labels (raytracing_system.cpp:1024, 1099, 1207). */
typedef struct CapStats
{
    uint64_t rays_primary;   /* rays actually traced since the last cap_stats_reset */
    uint64_t rays_extension;
    uint64_t rays_shadow;
    uint64_t shaded_vertices;
    uint64_t frames;
    double   ms_total;          /* GPU time of cap_render calls (hipEvent, context stream) */
    double   ms_primary;        /* "RaytracePrimaryVisibility" (fused small-scene path: + shading of the camera vertex) */
    double   ms_trace_closest;  /* extension-ray traversal ("RT Indirect diffuse"; fused small-scene path: + shading) */
    double   ms_trace_any;      /* shadow-ray traversal */
    double   ms_shade;          /* shading / BSDF sampling / compaction */
    double   ms_resolve;        /* radiance accumulate */
    uint64_t launches_trace_closest;
    uint64_t launches_trace_any;
    uint64_t launches_shade;   /* 0 when the small-scene path fuses shading into the closest-hit kernel */
    uint64_t rays_extension_bounce0; /* extension / shadow rays emitted by the bounce-0 (camera-vertex) kernel */
    uint64_t rays_shadow_bounce0;
    uint64_t guard_shade;     /* malformed queue entries caught by the kernels' bounds guards: always 0 in a correct run */
    uint64_t guard_trace_any;
    uint64_t guard_last;      /* (queue index or bounce) << 32 | path id of the last offender */
    double   ms_post;         /* reconstruction chain, "Spatial gather" .. "TAA" (cap_post_frame), always timed */
    uint64_t post_frames;
    /* The reference's remaining timestamp labels (gui_system.cpp:94-104 lists what AllocateTimestampQueryPair named). */
    double   ms_direct;       /* "RT Direct lighting": shading of the camera vertex + its shadow rays (part of ms_primary on the fused
                                 small-scene path, of ms_shade / ms_trace_any otherwise; CAP_RENDER_STAGE_TIMERS) */
    double   ms_post_pass[5]; /* "Spatial gather", "Temporal upscale", "EAW", "Combine illumination", "TAA" (sum = ms_post) */
    /* Shadow rays that were stored as an entry before being traced: in the shadow queue for an any-hit launch, or -- small-scene path,
     * bounces >= 1 -- in the per-wave ring of the kernel that generated them (origin in LDS, the 16-B contribution in the wave's slice
     * of the shadow queue's memory; bench.py prices either kind at 32 B: 16 + 16 written for a queue entry, 16 written + 16 read back
     * for a ring entry).  rays_shadow counts every shadow ray traced; on the small-scene path the generating kernel answers a shadow
     * ray itself when it can (reference model: a probe against the most likely occluder; EXT model: the whole test), and only the
     * rest become entries. */
    uint64_t shadow_entries;
    uint64_t shadow_entries_bounce0;
    /* Waves whose queue append ran past the capacity of their class's sub-queue: always 0 in a correct run (a path keeps the class
     * it got at bounce 0, so a sub-queue cannot receive more entries than the class has paths).  The entries beyond the capacity are
     * NOT stored -- nothing is written out of bounds -- and their paths are lost; guard_last then holds the two counter values the
     * offending wave saw (extension << 32 | shadow).  cap_debug_set(CAP_DEBUG_QUEUE_CAPACITY_DIV) provokes it for the tests. */
    uint64_t guard_append;
    /* cap_render calls that wanted two batch lanes (tree path, see cap_render) and ran on one because the second working set could not
     * be allocated: same image, ~7 % less throughput.  A (paths, bounces) size that has failed is not tried again until
     * cap_scene_upload / cap_set_resolution / cap_set_shard / cap_set_batch_paths change what is needed, the device reports enough free
     * memory for it (memory can come back without a call on this context), or every 32nd call. */
    uint64_t lane1_dropped;
} CapStats;

typedef struct CapBvhInfo
{
    uint32_t triangle_count;
    uint32_t node_count;  /* internal nodes (triangle_count - 1, or 0) */
    uint32_t max_depth;
    uint32_t stack_entries; /* per-lane LDS traversal stack the trace kernels were specialised for */
    float    bounds_lo[3];
    float    bounds_hi[3];
    double   build_ms;
} CapBvhInfo;

const char* cap_last_error(void);
/* number of HIP devices visible (0 without a GPU); never fails */
int cap_device_count(void);

/* Replaces Dx12 device/queue creation (dx12/dx12.cpp:165-235) + RaytracingSystem ctor (raytracing_system.cpp:182).
 * hip_stream: an existing hipStream_t to run on (e.g. the caller's torch stream), or NULL to create one. */
int  cap_ctx_create(int device_id, void* hip_stream, CapContext** out_ctx);
void cap_ctx_destroy(CapContext* ctx);

/* GeometryStorage upload, CreateGeometryStorage (asset_load_system.cpp:162-255): pooled positions/normals
 * (3 floats per vertex), texcoords (2 per vertex), mesh-local uint32 indices, 32-byte mesh descriptors. */
int cap_scene_upload(CapContext* ctx, const float* positions, const float* normals, const float* texcoords,
                     const uint32_t* indices, const CapMeshDesc* meshes, uint32_t vertex_count, uint32_t index_count,
                     uint32_t mesh_count);
/* TextureSystem::LoadTexture upload half (texture_system.cpp:58-118): RGBA8, row 0 first. rgba8 == NULL
 * installs the reference's "missing texture" 1x1 zero texel (texture_system.cpp:47-56). */
int cap_texture_upload(CapContext* ctx, uint32_t index, const uint8_t* rgba8, uint32_t width, uint32_t height);
/* Blue-noise texture load (raytracing_system.cpp:642-646): 256x256 RGBA8; only R,G are read (sampling.h:13-23). */
int cap_bluenoise_upload(CapContext* ctx, const uint8_t* rgba8_256x256);
/* EXT */
int cap_materials_upload(CapContext* ctx, const CapMaterial* materials, uint32_t mesh_count);

/* Replaces BLASSystem::BuildBLAS + TLASSystem::BuildTLAS (blas_system.cpp:14-67, tlas_system.cpp:11-73):
 * explicit on-device LBVH over all meshes; (instance, primitive) ids are kept per triangle. */
int cap_bvh_build(CapContext* ctx);
/* How cap_bvh_build builds the tree (the hits are the same either way; every build ends with the collapse into the compressed
 * 8-wide view).  LBVH: Morton hierarchy on the device, ~4 ms for 262 k triangles.  PLOC: agglomerative clustering on the device
 * with a surface-area distance (ploc.hip), ~3 ms, shadow and camera rays as fast as with the SAH tree -- the build for scenes
 * that change.  SAH: binned surface-area heuristic on the host, ~0.15 s for 262 k triangles, fastest traversal -- the counterpart of
 * D3D12_RAYTRACING_ACCELERATION_STRUCTURE_BUILD_FLAG_PREFER_FAST_TRACE, which the reference asks for (blas_system.cpp:42-47) while
 * building only once (tlas_system.cpp:111-121).  AUTO: PLOC above 64 triangles (trace times within 1 % of the SAH tree's; smaller
 * scenes are traced exhaustively and get the Morton hierarchy). */
typedef enum CapBvhBuild
{
    CAP_BVH_BUILD_AUTO = 0,
    CAP_BVH_BUILD_LBVH = 1,
    CAP_BVH_BUILD_SAH  = 2,
    CAP_BVH_BUILD_PLOC = 3, /* on the device: agglomerative clustering over the Morton order with a surface-area distance */
    CAP_BVH_BUILD_SAH_DEVICE = 4 /* on the device: binned surface-area splits from the root down to small segments, the clustering
                                  * inside those -- the tree quality the reference requests (PREFER_FAST_TRACE, blas_system.cpp:44) */
} CapBvhBuild;
int cap_set_bvh_build(CapContext* ctx, uint32_t mode);
int cap_bvh_info(CapContext* ctx, CapBvhInfo* out);
/* Debug/test readback: nodes = node_count * 16 floats (see DESIGN.md "BVH node"), leaf_triangles = triangle ids in leaf order. */
int cap_bvh_readback(CapContext* ctx, float* nodes, uint32_t* leaf_triangles);
/* The compressed 8-wide view the extension- and shadow-ray kernels walk (capsaicin_amd/csrc/cap_wide.h): nodes = 20 uint32 per
 * node (info[0] nodes; call with nodes = NULL first), tri_src = leaf position per wide-order triangle record (triangle_count
 * entries), info = {node count, depth, leading top-level nodes}.  Built by cap_bvh_build: on the device when the device built
 * the binary tree (CAP_BVH_BUILD_LBVH), on the host from the host's SAH tree.  For tests and tools. */
int cap_bvh_wide_readback(CapContext* ctx, uint32_t* nodes, uint32_t* tri_src, uint32_t* info);

/* CameraSystem::Run upload (camera_system.cpp:89-131). sensor_size is used as given (the caller applies
 * AdjustCameraAspectBasedOnWindow, camera_system.cpp:10-17). */
int cap_camera_set(CapContext* ctx, const CapCameraData* camera);
/* CameraComponent::prev_camera_buffer (camera_system.cpp:89-131; bound as g_prev_camera, raytracing_system.cpp:1227-1228) */
int cap_prev_camera_set(CapContext* ctx, const CapCameraData* camera);
/* RenderSystem::window_width/height (render_system.h) */
int cap_set_resolution(CapContext* ctx, uint32_t width, uint32_t height);
/* Screen-tile sharding: this context renders the 8x8 tiles t with t % shard_count == shard_index. */
int cap_set_shard(CapContext* ctx, uint32_t shard_index, uint32_t shard_count);
/* Upper bound of (frame, pixel) paths kept in flight per batch.  0 = default: 128 Mi paths (~28 GB of queues and planes per working
 * set; the tree path keeps two) when a call has more than 64 Mi to render and the device reports room for three such working sets,
 * 64 Mi otherwise; at most 64 frames per batch either way.  Results do not depend on it. */
int cap_set_batch_paths(CapContext* ctx, uint64_t max_paths);
/* Test hooks (no reference counterpart; never needed by a host program).  CAP_DEBUG_QUEUE_CAPACITY_DIV: the sub-queues of the next
 * renders get 1 / value of the capacity they need (value 1 = normal), so that the kernels' append guard (CapStats::guard_append) can
 * be seen to fire -- entries beyond a sub-queue are dropped and counted, never written. */
enum
{
    CAP_DEBUG_QUEUE_CAPACITY_DIV = 1,
    /* The extension- and shadow-ray kernels walk the compressed 8-wide view of the tree while its depth fits their per-lane stacks
     * (LDS part + spill slice: 21 levels) and fall back to the binary tree's kernels beyond.  WIDE_DEPTH_LIMIT (0 = none) lowers that
     * bound so that the fallback can be exercised on ordinary scenes; WIDE_IN_USE (read-only) says which kernels the next render
     * takes. */
    CAP_DEBUG_WIDE_DEPTH_LIMIT   = 2,
    CAP_DEBUG_WIDE_IN_USE        = 3,
    /* FAIL_LANE1 (0 / 1): the allocation of cap_render's second batch lane fails (what running out of HBM does), so that the one-lane
     * fallback and CapStats::lane1_dropped can be tested; LANES_USED (read-only): the lanes the last cap_render ran on (1 or 2). */
    CAP_DEBUG_FAIL_LANE1         = 4,
    CAP_DEBUG_LANES_USED         = 5,
    /* Canary behind the last queue class (the append guard's direct proof).  CANARY_FILL (set, after a first cap_render has allocated
     * them): every entry of the context's extension-queue planes takes one marker word.  CANARY_BEHIND (get): entries BEHIND the 64
     * sub-queues of the last render (index >= 64 x its sub-queue capacity) that no longer hold the marker -- 0 unless something wrote
     * past the last class; CANARY_USED (get): the same count inside the sub-queues (> 0 after any render: the check can see writes). */
    CAP_DEBUG_QUEUE_CANARY_FILL   = 6,
    CAP_DEBUG_QUEUE_CANARY_BEHIND = 7,
    CAP_DEBUG_QUEUE_CANARY_USED   = 8,
    /* SELFTEST_DIV (get, ~0.5 s): the reconstruction chain's exact mode evaluates its per-tap divisions without the scaling steps of
     * the compiler's IEEE expansion where the operands are in a range that never triggers them (post.hip div_unscaled); this runs both
     * forms on the device -- every normal float through log2, 2^30 operand pairs over the whole range -- and returns the number of
     * results that differ in any bit: 0. */
    CAP_DEBUG_SELFTEST_DIV        = 9,
    /* NEE_PAIRS (get): fan pairs the EXT model's next-event rays test on the small-scene path << 32 | fan pairs of the scene.  Pairs that
     * support the scene's convex hull with every light at a safe distance inside cannot occlude a segment between a scene point and a
     * light point under the intersection contract (rule and error bound: context.hip update_nee_pairs) and are left out. */
    CAP_DEBUG_NEE_PAIRS           = 10,
    /* A/B and diagnostic switches of the build and render paths (which kernels trace the camera and the shadow rays, one or two batch
     * lanes, the builders' parameters ...): ONE table per context, key = SWITCH_BASE + cap_debug_switch_index("CAP_..."), the names being
     * the environment variables that fill the table once, at cap_ctx_create (tools set those around a whole process; nothing else in the
     * library reads the environment except CAP_RCCL_LIBRARY).  value: the switch's number (flags: 1 / 0), ~0 = the product's own
     * choice again.  A switch is read by the next cap_bvh_build / cap_render.  Every switch selects between paths that give the same
     * image: they exist for measurements and for tests that exercise a path the product would not take on a small scene. */
    CAP_DEBUG_SWITCH_BASE         = 64
};
int cap_debug_set(CapContext* ctx, uint32_t key, uint64_t value);
int cap_debug_get(CapContext* ctx, uint32_t key, uint64_t* value);
/* Index of a switch by its name ("CAP_NO_WIDE8", "CAP_PRIMARY_WIDE", ...; capsaicin_amd/csrc/cap_kernels.h CapSwitch), -1 if unknown. */
int cap_debug_switch_index(const char* name);
/* Traversal strategy of the trace kernels (same hits either way): AUTO picks EXHAUSTIVE for scenes of at most 64
 * triangles (wave-uniform test of every triangle, no stack) and STACK (LBVH + per-lane LDS stack) otherwise. */
typedef enum CapTraversal
{
    CAP_TRAVERSAL_AUTO       = 0,
    CAP_TRAVERSAL_STACK      = 1,
    CAP_TRAVERSAL_EXHAUSTIVE = 2 /* refused above 4096 triangles */
} CapTraversal;
int cap_set_traversal(CapContext* ctx, uint32_t mode);

/* RaytracingSystem::Run ray passes (raytracing_system.cpp:266-292) for frame_count = frame_begin ..
 * frame_begin + n_frames - 1, each frame's COMBINED added to the accumulation buffer in frame order.
 * num_bounces == SettingsComponent::num_diffuse_bounces (gui_system.h:39).  Asynchronous on the context
 * stream; cap_readback / cap_stats_get / cap_sync wait for it. */
int cap_render(CapContext* ctx, uint32_t frame_begin, uint32_t n_frames, uint32_t num_bounces, uint32_t flags);
int cap_accum_reset(CapContext* ctx);
/* Resume of a long accumulation (SURVEY.md 5 "checkpoint": dump accumulation + sample index): sum_rgba = width*height*4 floats as
 * cap_readback(CAP_BUF_ACCUM_SUM) returned them (running sums in .xyz, frames in .w), frames = how many frames they hold; the next
 * cap_render(frames, n, ..) continues the sum exactly where the dumped one stopped -- the additions are the same, in the same frame
 * order, so the result is bit-identical to an uninterrupted render.  A sharded context takes its own tiles of the image. */
int cap_accum_import(CapContext* ctx, const float* sum_rgba, uint64_t frames);
int cap_sync(CapContext* ctx);

/* dst: width*height*4 floats (host), row 0 = pixel row 0.  Pixels outside this context's shard read 0. */
int cap_readback(CapContext* ctx, CapBufferKind kind, float* dst);

int cap_stats_get(CapContext* ctx, CapStats* out);
int cap_stats_reset(CapContext* ctx);

/* ---- multi-GPU tile exchange (one gather of tile radiance at frame end) ---- */
/* floats in this context's tile-ordered radiance buffer: max_tiles_per_shard * 64 * 4 (same on every shard) */
int cap_tile_buffer_floats(CapContext* ctx, size_t* out_floats);
/* writes ACCUM_MEAN of the local tiles, tile order, into a DEVICE buffer of cap_tile_buffer_floats floats */
int cap_resolve_tiles(CapContext* ctx, float* device_dst);
/* device_src: shard_count tile buffers back to back (the gather result); device_image: width*height*4 floats */
int cap_assemble_tiles(CapContext* ctx, const float* device_src, uint32_t shard_count, float* device_image);

/* ---- the exchange itself: ONE gather of tile radiance to rank 0 per frame, over RCCL (xGMI) ----
 * The reference is single-GPU (dx12.cpp:13-25, 196-234); this is the north star's "frames shard by screen tile across the GPUs of
 * one node with a single RCCL gather of tile radiance at frame end".  Each context renders the shard cap_set_shard gave it
 * (shard index == rank, shard count == ranks).  RCCL is loaded on first use (dlopen "librccl.so.1", or CAP_RCCL_LIBRARY); a copy
 * the process already has loaded is shared.  cap_comm_gather_frame* resolve the tiles (as cap_resolve_tiles), gather them to rank
 * 0 with ncclGather on the contexts' streams and assemble the row-major image there (as cap_assemble_tiles): asynchronous,
 * ordered behind the render on each stream. */
#define CAP_COMM_ID_BYTES 128
/* one process per GPU: rank 0 makes the id, the host program carries it to the other ranks (file, MPI, a key-value store) */
int cap_comm_unique_id(uint8_t* id_128_bytes);
int cap_comm_init_rank(CapContext* ctx, const uint8_t* id_128_bytes, uint32_t rank, uint32_t nranks); /* collective */
int cap_comm_gather_frame(CapContext* ctx);                                                            /* collective */
/* one process driving n GPUs: ctxs[i] renders shard i of n; contexts on pairwise distinct devices get an RCCL communicator
 * (ncclCommInitAll), contexts that all share one device exchange by device copies */
int cap_comm_init_all(CapContext* const* ctxs, uint32_t n);
int cap_comm_gather_frame_all(CapContext* const* ctxs, uint32_t n);
/* rank 0 after a gather: the assembled frame, width*height*4 floats (mean radiance; .w = frames), on the device / on the host */
int cap_comm_image(CapContext* ctx, float** device_image);
int cap_comm_readback(CapContext* ctx, float* dst);
int cap_comm_info(CapContext* ctx, uint32_t* rank, uint32_t* size, uint32_t* uses_rccl);
int cap_comm_destroy(CapContext* ctx); /* also done by cap_ctx_destroy */
/* Error path: gives the communicator up WITHOUT waiting for the stream (ncclCommAbort) -- for a rank whose peers failed before
 * entering a collective this rank has already queued, where cap_comm_destroy's stream synchronisation would never return.  The
 * communicator's asynchronous error state is polled once per frame by cap_comm_gather_frame* (ncclCommGetAsyncError). */
int cap_comm_abort(CapContext* ctx);

/* ---- reconstruction chain ----
 * The passes RaytracingSystem::Run records after the ray passes (raytracing_system.cpp:294-317):
 * SpatialGather (cpp:1541-1604) -> IntegrateTemporally (cpp:1283-1342) -> Denoise = BlurDisocclusion + 2|4 a-trous blurs
 * (cpp:1437-1538) -> CombineIllumination (cpp:1400-1435) -> ApplyTAA (cpp:1344-1398), full-resolution configuration.
 * Settings = the SettingsComponent fields those passes read (gui_system.h:20-37, defaults in comments). */
typedef struct CapPostSettings
{
    int32_t gather;                    /* true  */
    int32_t denoise;                   /* true  */
    int32_t eaw5;                      /* true  */
    float   eaw_normal_sigma;          /* 128   */
    float   eaw_depth_sigma;           /* 3     */
    float   eaw_luma_sigma;            /* 3     */
    float   gather_normal_sigma;       /* 64    */
    float   gather_depth_sigma;        /* 2     */
    float   gather_luma_sigma;         /* 3     */
    float   temporal_upscale_feedback; /* 0.975 */
    float   taa_feedback;              /* 0.9   */
    int32_t lowres_indirect;           /* false: RaytracingOptions::lowres_indirect, UPSCALE2X in Gather and Accumulate
                                          (spatial_gather.hlsl:36-46, temporal_accumulation.hlsl:228-235, 307-313) */
    /* Every field from here on reads 0 as the reference's default, so that a caller who fills the struct positionally up to
     * lowres_indirect (the round-1 form) -- or memsets it and sets what it knows -- runs the reference's default configuration. */
    int32_t disable_variance;          /* false: NOT RaytracingOptions::use_variance (raytracing_system.h:25, default true): the
                                          USE_VARIANCE define of eaw_blur.hlsl:68,114,127,162 (raytracing_system.cpp:669-673).  Set:
                                          no luma edge-stopping and no a-trous kernel weights in Blur, variance channel 0 */
    int32_t fast_weights;              /* false.  Not a reference option: evaluates the edge-stopping weights with the hardware's
                                          v_exp_f32 / v_log_f32 / v_rcp_f32 instead of the arithmetic contract's polynomials and IEEE
                                          divisions, and (round 5) the VALUES of IntegrateTemporally and TAA with the same instructions
                                          and the centre tap of their bicubic resamples -- what those two passes DECIDE (reprojection,
                                          disocclusion test, static / moving) stays the exact arithmetic on the same G-buffer, so both
                                          modes reset and blend at the same pixels.  The exact mode (0) is bit-identical to the oracle; this one is held to a stated
                                          tolerance against it over a multi-frame sequence (tests/test_post_gpu.py), per colour channel
                                          with e = |fast - exact| / (|exact| + 1e-3): median e <= 2e-5, 99 % of the channels
                                          e <= 4e-3, every channel e <= 3e-2 (TAA's variance clipping amplifies in flat regions) */
    int32_t output;                    /* 0: SettingsComponent::output (gui_system.h:11-17, 38), passed to CombineIllumination as
                                          `type` (raytracing_system.cpp:1415; combine_illumination.hlsl:26-40): CAP_OUTPUT_COMBINED
                                          indirect * albedo + direct, CAP_OUTPUT_DIRECT, CAP_OUTPUT_INDIRECT (the denoised indirect
                                          term, w = 1), CAP_OUTPUT_VARIANCE (its variance channel: indirect.www after the last blur) */
} CapPostSettings;
enum
{
    CAP_OUTPUT_COMBINED = 0, /* kCombined */
    CAP_OUTPUT_DIRECT   = 1, /* kDirect   */
    CAP_OUTPUT_INDIRECT = 2, /* kIndirect */
    CAP_OUTPUT_VARIANCE = 3  /* kVariance */
};
/* The reference's default configuration (gui_system.h:20-40, raytracing_system.h:22-27): gather, denoise, eaw5 on; sigmas 128 / 3 / 3
 * and 64 / 2 / 3; feedbacks 0.975 / 0.9; every later field 0. */
void cap_post_settings_default(CapPostSettings* out);
/* Runs the chain on the planes of the last frame rendered with CAP_RENDER_AOV (frame_count = that frame's index; the
 * camera is the one set for it; prev_camera = the previous frame's, CameraComponent/prev_camera of
 * temporal_accumulation.hlsl:10-11) and keeps the histories for the next call.  Needs an unsharded context: the stencils read
 * across tile borders.  Asynchronous on the context stream.  The result (current_frame_output(), cpp:320-324) is read with
 * cap_post_readback. */
int cap_post_frame(CapContext* ctx, const CapPostSettings* settings, uint32_t frame_count, const CapCameraData* prev_camera);
/* Sharded contexts: the chain runs on ONE rank on the gathered ray-pass outputs (the note on the post-process chain in the
 * multi-GPU design: 5x5 .. 7x7 neighbourhoods with strides up to 14 px cross every tile border).  Per frame:
 *   every rank   cap_render(.., CAP_RENDER_AOV); cap_resolve_aov_tiles(ctx, buf)            buf: cap_aov_tile_buffer_floats floats
 *   one gather of the rank buffers to the root (rank-major, like cap_resolve_tiles / cap_assemble_tiles)
 *   root         cap_post_frame_gathered(ctx, settings, frame_count, prev_camera, gathered, shard_count)
 * The buffer holds the four planes the chain reads (indirect, direct, albedo, normal/depth of the AOV frame), plane-major, each
 * in tile order.  The root's context must have the same resolution, camera and shard_count; its own render outputs are not
 * used (with settings.lowres_indirect every rank renders with CAP_RENDER_LOWRES_INDIRECT and frame_count selects the 2x2
 * interleave offset, as in cap_post_frame). */
int cap_aov_tile_buffer_floats(CapContext* ctx, size_t* out_floats);
int cap_resolve_aov_tiles(CapContext* ctx, float* device_dst);
int cap_post_frame_gathered(CapContext* ctx, const CapPostSettings* settings, uint32_t frame_count, const CapCameraData* prev_camera,
                            const float* device_gathered, uint32_t shard_count);
/* CAP_RENDER_GBUFFER_FEEDBACK on sharded contexts: the next frame's indirect pass reads the chain's output and the normal/depth
 * image of this frame (rt_indirect.hlsl:116-145), which only the root has.  After the chain of frame f the root exports the two
 * images (cap_feedback_buffer_floats floats: width*height*4 each, output first), ONE broadcast carries them to the other ranks,
 * and those import them before cap_render(f + 1, .., CAP_RENDER_GBUFFER_FEEDBACK).  Frame 0 needs nothing (cleared histories
 * everywhere).
 * Layout of the buffer (a contract between builds: ranks exchange it): plane 0 = the chain's output of frame f,
 * current_frame_output() (RGBA as cap_post_readback returns it); plane 1 = frame f's normal/depth image as the CHAIN keeps it for
 * the next frame -- the DECODED unit normal in .xyz (OctDecode of gbuffer_normal_depth.xy) and the raw depth in .w -- not the
 * oct-encoded G-buffer plane of CAP_BUF_NORMAL_DEPTH (that was its content until round 3).  The indirect pass's feedback branch and
 * Accumulate read only .w; both planes are written and read by cap_feedback_export / _import only, as a pair
 * (tests/test_post_gpu.py::test_gbuffer_feedback_on_shards round-trips them across every frame of a sequence). */
int cap_feedback_buffer_floats(CapContext* ctx, size_t* out_floats);
int cap_feedback_export(CapContext* ctx, float* device_dst);
int cap_feedback_import(CapContext* ctx, const float* device_src, uint32_t frame_count);
/* zero-fills the histories (a new sequence; also implied by cap_set_resolution) */
int cap_post_reset(CapContext* ctx);
/* dst: width*height*4 floats (host) */
int cap_post_readback(CapContext* ctx, float* dst);

#ifdef __cplusplus
}
#endif
#endif /* CAPSAICIN_HIP_H */
