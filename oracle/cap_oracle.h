/*
 * cap_oracle.h — C interface of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is a scalar fp32 restatement of the reference renderer's hot path
 * (/root/reference/src/core/shaders/rt_primary_visibility.hlsl, rt_direct_lighting.hlsl,
 * rt_indirect.hlsl and the headers camera.h, sampling.h, lighting.h, shading.h, scene.h,
 * math_functions.h, data_payload.h).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product library (capsaicin_amd/csrc) never links,
 * includes or calls anything in this directory.
 *
 * PARITY UNPINNED at the TraceRay boundary: the reference delegates BVH build, traversal and
 * ray/triangle intersection to the closed-source D3D12/DXR driver (blas_system.cpp:65,
 * tlas_system.cpp:72, rt_primary_visibility.hlsl:44) and ships no tests or golden vectors, and it
 * cannot be built here (Windows + D3D12 + missing dxcompiler.dll).  What IS pinned: the pure
 * functions below are checked against the hand-evaluated known answers of SURVEY.md 8c
 * (tests/golden/kat.json) and against the reference's own data assets (blue-noise texels, parsed
 * Cornell box counts/bounds).
 */
#ifndef CAP_ORACLE_H
#define CAP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* camera_system.h:16-31 / data_payload.h:7-18 — 72 byte POD. */
typedef struct OracleCamera
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
} OracleCamera;

/* asset_load_system.h:29-39 / data_payload.h:20-30 — 32 byte POD. */
typedef struct OracleMesh
{
    uint32_t vertex_count;
    uint32_t first_vertex_offset;
    uint32_t index_count;
    uint32_t first_index_offset;
    uint32_t index;
    uint32_t texture_index; /* ~0u = untextured */
    uint32_t padding[2];
} OracleMesh;

typedef struct OracleTexture
{
    const uint8_t* rgba8; /* width*height*4, row 0 first */
    uint32_t       width;
    uint32_t       height;
} OracleTexture;

/* EXT (no reference counterpart): per-mesh material used only when
 * ORACLE_FLAG_EXT_MATERIALS is set.  kd replaces the texture/0.75 lookup of scene.h:52-61. */
typedef struct OracleMaterial
{
    float kd[3];
    float roughness; /* GGX alpha = roughness^2; >= 1 means pure Lambert */
    float ks[3];
    float pad0;
    float ke[3]; /* emitted radiance */
    float pad1;
} OracleMaterial;

/* GeometryStorage layout of asset_load_system.h:16-27 (pooled, indices mesh-local). */
typedef struct OracleScene
{
    const float*          positions; /* 3 * vertex_count */
    const float*          normals;   /* 3 * vertex_count */
    const float*          texcoords; /* 2 * vertex_count */
    const uint32_t*       indices;   /* index_count */
    const OracleMesh*     meshes;
    uint32_t              mesh_count;
    uint32_t              vertex_count;
    uint32_t              index_count;
    const OracleTexture*  textures;
    uint32_t              texture_count;
    const OracleMaterial* materials; /* mesh_count entries or NULL */
} OracleScene;

enum
{
    ORACLE_FLAG_USE_BVH       = 1u << 0, /* CPU BVH instead of brute force (same hits by construction) */
    ORACLE_FLAG_EXT_MATERIALS = 1u << 1, /* EXT: per-mesh kd / GGX / emission + next-event estimation */
    /* RaytracingOptions::lowres_indirect (raytracing_system.h:24; LOWRES_INDIRECT, rt_indirect.hlsl:53-59): the indirect pass
     * runs on the (W/2, H/2) grid, 2x2-interleaved over four frames; needs even W and H; fills indirect_lowres */
    ORACLE_FLAG_LOWRES_INDIRECT = 1u << 2,
};

typedef struct OracleFrameOutputs
{
    /* all W*H*4 floats, row 0 = pixel row 0; any pointer may be NULL */
    float* gbuffer_geo;   /* (u, v, asfloat(instance), asfloat(prim))  rt_primary_visibility.hlsl:46 */
    float* direct;        /* rt_direct_lighting.hlsl:53,68,78 */
    float* albedo;        /* rt_direct_lighting.hlsl:54,69,79 */
    float* normal_depth;  /* (oct(n).xy, instance, |cam-p|) rt_direct_lighting.hlsl:80 */
    float* indirect;      /* rt_indirect.hlsl:176 */
    float* combined;      /* indirect*albedo + direct, combine_illumination.hlsl:29 */
    uint64_t rays[3];     /* primary, extension, shadow rays actually traced */
    float* indirect_lowres; /* ORACLE_FLAG_LOWRES_INDIRECT: (W/2)*(H/2)*4 floats, g_output_indirect of the half-res pass; or NULL */
} OracleFrameOutputs;

void* oracle_scene_create(const OracleScene* scene);
void  oracle_scene_destroy(void* h);

/* One frame (= one sample per pixel) of the reference's three ray passes.  Returns 0 on success. */
int oracle_render_frame(void* scene, const OracleCamera* cam, const uint8_t* bluenoise_rgba8, uint32_t width,
                        uint32_t height, uint32_t frame_count, uint32_t num_bounces, uint32_t flags,
                        uint32_t num_threads, OracleFrameOutputs* out);

/* The same frame with the G-buffer feedback branch on (RaytracingOptions::gbuffer_feedback, raytracing_system.h:26;
 * rt_indirect.hlsl:116-145): prev_normal_depth and color_history are the previous frame's gbuffer_normal_depth and
 * reconstruction output (W*H*4 floats each, zero-filled before the first frame).  Reference shading model only. */
/* rows [y0, y1) only; pixels outside keep what the output buffers held, rays[] counts the rendered rows */
int oracle_render_frame_rows(void* scene, const OracleCamera* cam, const uint8_t* bluenoise_rgba8, uint32_t width, uint32_t height,
                             uint32_t frame_count, uint32_t num_bounces, uint32_t flags, uint32_t num_threads, uint32_t y0, uint32_t y1,
                             OracleFrameOutputs* out);
int oracle_render_frame_feedback(void* scene, const OracleCamera* cam, const OracleCamera* prev_cam, const uint8_t* bluenoise_rgba8,
                                 uint32_t width, uint32_t height, uint32_t frame_count, uint32_t num_bounces, uint32_t flags,
                                 uint32_t num_threads, const float* prev_normal_depth, const float* color_history,
                                 OracleFrameOutputs* out);

/* accum[i] += combined[i] for frames frame_begin .. frame_begin+n_frames-1 in increasing order
 * (plain fp32 running sum).  accum is W*H*4 floats, caller-zeroed. */
int oracle_render_accumulate(void* scene, const OracleCamera* cam, const uint8_t* bluenoise_rgba8, uint32_t width,
                             uint32_t height, uint32_t frame_begin, uint32_t n_frames, uint32_t num_bounces,
                             uint32_t flags, uint32_t num_threads, float* accum, uint64_t rays[3]);

/* ---- reconstruction chain, cap_oracle_post.cpp ---- */
/* SettingsComponent subset, gui_system.h:20-37 (defaults in comments) */
typedef struct OraclePostSettings
{
    int   gather;                    /* true  */
    int   denoise;                   /* true  */
    int   eaw5;                      /* true  */
    float eaw_normal_sigma;          /* 128   */
    float eaw_depth_sigma;           /* 3     */
    float eaw_luma_sigma;            /* 3     */
    float gather_normal_sigma;       /* 64    */
    float gather_depth_sigma;        /* 2     */
    float gather_luma_sigma;         /* 3     */
    float temporal_upscale_feedback; /* 0.975 */
    float taa_feedback;              /* 0.9   */
    int   lowres_indirect;           /* false: RaytracingOptions::lowres_indirect (UPSCALE2X in Gather and Accumulate); the
                                        `indirect` argument of oracle_post_frame is then the (W/2)*(H/2) image */
    int   use_variance;              /* true: RaytracingOptions::use_variance (raytracing_system.h:25) = the USE_VARIANCE define of
                                        eaw_blur.hlsl (raytracing_system.cpp:669-673); CALCULATE_VARIANCE (cpp:618-622) is read by no shader */
    int   output;                    /* 0: SettingsComponent::output (gui_system.h:11-17, 38) = CombineIllumination's `type`
                                        (raytracing_system.cpp:1415; combine_illumination.hlsl:26-40): 0 combined, 1 direct, 2 indirect, 3 variance */
} OraclePostSettings;

void* oracle_post_create(uint32_t width, uint32_t height);
void  oracle_post_destroy(void* chain);
/* Host threads the chain's passes split their rows over (default 1); the images do not depend on it. */
void  oracle_post_set_threads(int n);
/* One frame of Gather -> Accumulate -> BlurDisocclusion -> Blur x2|x4 -> Combine -> TAA on this frame's ray-pass outputs
 * (all W*H*4 floats); writes current_frame_output() (raytracing_system.cpp:320-324) to out and keeps the histories. */
int oracle_post_frame(void* chain, const OraclePostSettings* settings, uint32_t frame_count, const OracleCamera* camera,
                      const OracleCamera* prev_camera, const float* indirect, const float* direct, const float* albedo,
                      const float* normal_depth, float* out);
/* one pass on caller-supplied full-resolution images (see cap_oracle_post.cpp): 0 Gather, 1 Accumulate (arg = frame_count),
 * 2 BlurDisocclusion, 3 Blur (arg = stride), 4 TAA */
int oracle_post_pass(int pass, const OraclePostSettings* settings, uint32_t width, uint32_t height, uint32_t arg, const OracleCamera* camera,
                     const OracleCamera* prev_camera, const float* in0, const float* in1, const float* in2, const float* in3,
                     const float* in4, float* out0, float* out1);

/* ---- pure functions (known-answer tests) ---- */
void     oracle_halton23(uint32_t frame_count, float out[2]);                                   /* sampling.h:143-155 */
uint32_t oracle_wang_hash(uint32_t x, uint32_t y);                                              /* sampling.h:37-46 */
void     oracle_bluenoise4x4(const uint8_t* rgba8, uint32_t x, uint32_t y, uint32_t count, float out[2]); /* sampling.h:13-23 */
void     oracle_directional_light(uint32_t count, float dir[3], float intensity[3]);            /* lighting.h:20-33 */
void     oracle_primary_ray(const OracleCamera* cam, uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                            uint32_t frame_count, float origin[3], float dir[3]);               /* camera.h:39-63 */
void     oracle_map_to_hemisphere(const float s[2], const float n[3], float out[3]);            /* sampling.h:113-132, e = 1 */
void     oracle_sincos(float x, float* s, float* c);
float    oracle_pow22(float x);                                                                 /* scene.h:58 */
void     oracle_oct_encode(const float n[3], float out[2]);                                     /* math_functions.h:41-47 */
/* returns 1 and fills t,u,v when the ray hits the triangle with tmin < t < tmax */
int      oracle_intersect_triangle(const float o[3], const float d[3], float tmin, float tmax, const float v0[3],
                                   const float v1[3], const float v2[3], float* t, float* u, float* v);
void     oracle_sample_texture(const OracleTexture* tex, float u, float v, float out[3]);       /* scene.h:55-58 */

#ifdef __cplusplus
}
#endif
#endif
