// cap_device.h — device-side data layout shared by the kernels and the host context (DESIGN.md "Data layout in HBM").
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cap_leaf.h"
#include "cap_math.h"

namespace cap
{
constexpr uint32_t kInvalidId  = ~0u;  // data_payload.h:5
constexpr uint32_t kTileDim    = 8;    // one 8x8 screen tile == one 64-lane wavefront of primary rays
constexpr uint32_t kTilePixels = 64;
constexpr uint32_t kPidShift   = 26;   // path id = (frame slot << 26) | local pixel
constexpr uint32_t kPidMask    = (1u << kPidShift) - 1u;
constexpr uint32_t kMaxFrameSlots = 64;
constexpr uint32_t kBlock      = 256;  // threads per workgroup (4 waves, one per SIMD)
// Queue classes.  A device-scope atomic on ONE word saturates near 88 ops/us on MI355X (MI355X_MICROARCH.md "dequeue"),
// far below what one append per wave needs, so every queue is split into kQueueClasses independent sub-queues, each with
// its own counter on its own 128-byte line.  A path's class is fixed at bounce 0 (64-path chunk index mod kQueueClasses)
// and never changes, so sub-queue k can never hold more than the class-k paths: static capacity, no overflow handling.
constexpr uint32_t kQueueClasses  = 64;
// Which class a 64-path chunk of bounce 0 gets, and the inverse (the j-th chunk of class k).  Every aligned block of 64 chunks holds
// each class exactly once, so class k still receives ceil(chunks / 64) chunks -- the static capacity -- but the assignment ROTATES
// with the block index.  With the plain `chunk mod 64` of rounds 1-3 a class was tied to whatever `chunk mod 64` means on the screen:
// for a tile-sharded 4096 x 4096 frame (512 tile columns, shard s = tile columns s mod 8) that is ONE 8-pixel-wide column of the
// image per class for all its frames, the classes' path counts after the first bounce differ by what their columns look at, and a
// launch lasts as long as its longest class (waves never change class): 23.3 instead of 31.2 Grays/s for one rank's share of
// BASELINE configs[4] (docs/experiments.md (54)).  The rotation walks every class across all columns.
constexpr uint32_t kQueueClassBits = 6;
__host__ __device__ __forceinline__ uint32_t chunk_class(uint32_t chunk) { return (chunk + (chunk >> kQueueClassBits)) & (kQueueClasses - 1u); }
__host__ __device__ __forceinline__ uint32_t class_chunk(uint32_t j, uint32_t klass) { return (j << kQueueClassBits) | ((klass - j) & (kQueueClasses - 1u)); }
constexpr uint32_t kCounterStride = 32;  // uint32 words between two class counters (128 B)
// Words 2 and 3 of a class's counter line: shadow rays answered by the producer's probe, shaded vertices (flush_stats)
constexpr uint32_t kShadeRec      = 8;   // float4 per shading record
constexpr uint32_t kExhaustiveMax = 64;  // scenes up to this many triangles are traced exhaustively (kernels.hip)
constexpr int      kNoChild        = 0x7fffffff;  // unused slot of a wide node
// k_trace_any on the 8-wide view: 24 LDS words per lane = 12 (g_base, g_mask) pairs (24 KB per workgroup: six workgroups per CU)
constexpr uint32_t kWideLdsEntries = 24;
constexpr uint32_t kSpillEntries   = 24;  // per-thread stack words (12 pairs) kept in global memory behind the LDS part

// BVH node, 64 B = 4 x float4 (both children's boxes live in the parent, one fetch tests both):
//   q0 = (lo0.x lo0.y lo0.z hi0.x)  q1 = (hi0.y hi0.z lo1.x lo1.y)  q2 = (lo1.z hi1.x hi1.y hi1.z)
//   q3 = (child0, child1, tchild0, tchild1) as int bits; child >= 0: internal node index, child < 0: ~leaf (sorted triangle)
//        index.  tchild = what the traversal follows: the same, except that a subtree of <= kLeafMax triangles is one leaf,
//        ~(first sorted triangle | (count - 1) << kLeafCountShift)
// Intersection triangle, 64 B = 4 x float4, in leaf order (n = cross(e1, e2)):
//   t0 = (v0.x v0.y v0.z e1.x)  t1 = (e1.y e1.z e2.x e2.y)  t2 = (e2.z n.x n.y n.z)  t3 = (asfloat(global triangle id), -, -, -)
// Shading triangle, 128 B = kShadeRec x float4 (two 64-B sectors, nothing else to fetch per shaded vertex), in global triangle
// order (mesh order, then primitive order):
//   s0 = (p0, uv0.x) s1 = (p1, uv0.y) s2 = (p2, uv1.x) s3 = (n0, uv1.y) s4 = (n1, uv2.x) s5 = (n2, uv2.y)
//   s6 = (instance, primitive, texture index of the instance's mesh, -) as int bits     s7 unused
struct BvhDev
{
    const float4* nodes;
    uint32_t*     stack_spill; // kSpillEntries words per thread of the persistent grid: stack entries beyond the LDS part
    uint32_t      spill_threads;  // threads the spill area is sized for
    const float4* tris;        // 64-B intersection records in leaf order (.w of the 4th float4 = global triangle id)
    const float4* tris_by_id;  // the same records in global triangle id order (exhaustive small-scene kernels)
    // exhaustive loop order: fan pairs (triangles id, id + 1 sharing v0 and the edge v0->v2), 20 floats each
    //   (v0, e1, e2, e3, nA, nB, asfloat(id), 0), ascending id; then the unpaired triangles as 64-B records, ascending id
    const float4* fan_pairs;
    const float4* fan_singles;
    uint32_t      fan_pair_count, fan_single_count;
    // EXT model, next-event rays (round 6): the same pair records with the pairs that can never occlude a segment between a scene point
    // and a point of a light triangle moved behind the first fan_pair_nee_count -- see update_nee_pairs() in context.hip for the rule
    // and its exactness argument.  An occlusion test is an OR over the pairs, so their order is free there.
    const float4* fan_pairs_nee;
    uint32_t      fan_pair_nee_count;
    int32_t       root;       // 0, or ~0 for a single triangle
    uint32_t      tri_count;  // 0 -> every ray misses
    // compressed 8-wide view (cap_wide.h): 5 x float4 per node, breadth-first; 64-B intersection records in its own leaf order
    const float4* nodes8;
    const float4* tris8;
    uint32_t      wide8_ok;   // built, and its depth fits the pair stacks of the wide kernels (LDS part + spill slice)
    uint32_t      wide8_top;  // leading nodes a workgroup may stage in LDS (<= kWideTopNodes)
};

// One entry per texel, row-major: the RGBA8 words of (x, y), (x + 1, y), (x, y + 1), (x + 1, y + 1), indices wrapped -- what one
// bilinear WRAP sample reads, in one 16-byte load (4 x the memory of the plain image: 16 MB for a 1024 x 1024 texture).
struct TextureDev
{
    const uint4* quads;
    uint32_t     width, height;
};

// Per-frame constants evaluated on the host with cap_math.h (camera.h:41 jitter, lighting.h:20-33 light).
struct FrameConst
{
    float    jitter_x, jitter_y;
    uint32_t frame_count;
    uint32_t lowres_sel;  // LOWRES_INDIRECT (rt_indirect.hlsl:53-59): bit 2 = on, bit 1 = sp_offset.x, bit 0 = sp_offset.y
    float    light_dir[3];
    float    pad1;
    float    light_intensity[3];
    float    pad2;
};

struct CameraDev
{
    float position[3], focal_length;
    float right[3], sensor_x;
    float forward[3], sensor_y;
    float up[3], pad;
};

// Inputs of the G-buffer feedback branch (rt_indirect.hlsl:116-145): the previous frame's camera, normal/depth G-buffer and
// TAA'd output, row-major width*height images owned by the reconstruction chain.
struct FeedbackDev
{
    CameraDev     prev_cam;
    const float4* prev_normal_depth;
    const float4* color_history;
};

// Screen decomposition of one context (shard): local pixel index pl = local_tile * 64 + (y_in_tile * 8 + x_in_tile),
// global tile = local_tile * shard_count + shard_index, tiles are numbered row-major over the 8x8 tile grid.
struct ScreenDev
{
    uint32_t width, height;
    uint32_t tiles_x, tiles_y;
    uint32_t tile_count;        // tiles_x * tiles_y
    uint32_t shard_index, shard_count;
    uint32_t local_tiles;       // tiles owned by this shard
    uint32_t pixels_padded;     // Ppad = max_tiles_per_shard * 64 (identical on every shard)
};

__device__ __forceinline__ bool local_pixel_to_xy(const ScreenDev& sc, uint32_t pl, uint32_t& x, uint32_t& y)
{
    const uint32_t lt = pl >> 6, w = pl & 63u;
    const uint32_t gt = lt * sc.shard_count + sc.shard_index;
    const uint32_t ty = gt / sc.tiles_x, tx = gt - ty * sc.tiles_x;
    x = tx * kTileDim + (w & 7u);
    y = ty * kTileDim + (w >> 3);
    return gt < sc.tile_count && x < sc.width && y < sc.height;
}

// camera.h:39-63
__device__ __forceinline__ v3 primary_dir(const CameraDev& cam, const ScreenDev& sc, const FrameConst& fc, uint32_t x, uint32_t y)
{
    const float ix = ((float)x + fc.jitter_x) / (float)sc.width, iy = ((float)y + fc.jitter_y) / (float)sc.height;
    const float cx = (ix - 0.5f) * cam.sensor_x, cy = (iy - 0.5f) * cam.sensor_y;
    const v3    d  = mk3(fmaf(cy, cam.up[0], fmaf(cx, cam.right[0], cam.focal_length * cam.forward[0])),
                         fmaf(cy, cam.up[1], fmaf(cx, cam.right[1], cam.focal_length * cam.forward[1])),
                         fmaf(cy, cam.up[2], fmaf(cx, cam.right[2], cam.focal_length * cam.forward[2])));
    return normalize3(d);
}

// Wavefront queues, all SoA planes of float4 (16 B per lane per access, 1 KiB per wave instruction).
// Sub-queue k occupies entries [k * class_capacity, (k + 1) * class_capacity) and is counted by count[k * kCounterStride].
struct RayQueue
{
    float4*   org_tmin;  // (o.xyz, tmin)
    float4*   dir_tmax;  // (d.xyz, tmax)
    float4*   thr_pid;   // (throughput.xyz, asfloat(path id))
    uint32_t* count;     // kQueueClasses device counters, kCounterStride apart
    uint32_t  class_capacity;  // entries per class, a multiple of 64
    float4*   acc;       // ShadeArgs::inline_nee only: (radiance the path has gathered since bounce 1, -), else unused
};
// EXT model: (o.xyz, tmin) (d.xyz, tmax) (contribution.xyz, asfloat(path id)).  Reference model (directional light, constant
// tmin / tmax, lighting.h:39-47): (o.xyz, asfloat(path id)) in org_tmin, (contribution.xyz, -) in contrib_pid, dir_tmax unused.
struct ShadowQueue
{
    float4*   org_tmin;
    float4*   dir_tmax;
    float4*   contrib_pid;  // radiance added when unoccluded
    uint32_t* count;
    uint32_t  class_capacity;
};

// Consumers walk "chunk slots": slot cs = j * kQueueClasses + k is the j-th 64-entry chunk of class k (class-minor order, so
// the occupied slots of all classes come first and spread evenly over the waves of the persistent grid).
__device__ __forceinline__ bool queue_chunk(const uint32_t* count, uint32_t class_capacity, uint32_t cs, uint32_t lane, uint32_t& index,
                                            uint32_t& klass)
{
    klass                = cs % kQueueClasses;
    const uint32_t j     = cs / kQueueClasses;
    const uint32_t n     = count[klass * kCounterStride];
    const uint32_t local = j * 64u + lane;
    index                = klass * class_capacity + local;
    return local < n;
}

// CapMaterial as uploaded (include/capsaicin_hip.h), 48 bytes
struct MaterialDev
{
    float kd[3], roughness;
    float ks[3], pad0;
    float ke[3], pad1;
};

struct SceneDev
{
    const float4*     shade_tris;  // kShadeRec per triangle
    const uint4*      tri_ids;     // (instance, primitive, texture index of the instance's mesh, -) per global triangle
                                   // (tlas_system.cpp:40-58; the texture index spares the shading a dependent load)
    const uint32_t*   mesh_texture;  // texture index per mesh (MeshComponent::material_index)
    const TextureDev* textures;
    uint32_t          texture_count;
    const float2*     bluenoise;   // 256*256 (R,G)/255
    float             kd_untextured;  // pow(0.75, 2.2), scene.h:55-58
    // EXT shading model (no reference counterpart; DESIGN.md "EXT shading model")
    const float2*      bluenoise_ba;  // 256*256 (B,A)/255
    const MaterialDev* materials;     // one per mesh
    uint32_t           material_count;
    const uint32_t*    light_tris;    // emissive triangles (global ids) in triangle order
    const float*       light_cdf;     // float prefix sums of their areas
    uint32_t           light_count;
    float              light_area;
};

struct Planes
{
    float4* color;   // [frame slot][Ppad]  indirect radiance (rt_indirect.hlsl color)
    float4* direct;  // [frame slot][Ppad]
    float4* albedo;  // [frame slot][Ppad]
    float4* aov_geo;           // [Ppad] last frame only (CAP_RENDER_AOV)
    float4* aov_normal_depth;  // [Ppad]
};
}  // namespace cap
