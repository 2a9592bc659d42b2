// capsaicin.cpp — the reference's public API (src/core/src/capsaicin.cpp:20-103) on top of the C ABI.
//
// The reference wires six systems into a yecs World and orders them with Precede edges (capsaicin.cpp:38-45, 58-62):
//   AssetLoad -> BLAS -> TLAS -> Camera -> Raytracing -> Composite -> GUI -> Render.
// yecs is an empty submodule directory in the reference tree; this file keeps the same system boundaries and the same
// total order in a small fixed-order world (the reference serialises the systems anyway, capsaicin.cpp:38-40 TODO).
#include "capsaicin.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../include/capsaicin_hip.h"
#include "../../include/capsaicin_scene.h"

namespace capsaicin
{
namespace
{
void info(const std::string& s) { std::fprintf(stderr, "[info] %s\n", s.c_str()); }
void warn(const std::string& s) { std::fprintf(stderr, "[warning] %s\n", s.c_str()); }
[[noreturn]] void error_throw(const std::string& s)
{
    std::fprintf(stderr, "[error] %s\n", s.c_str());
    throw std::runtime_error(s);  // dx12/common.h:17-32: log, then throw
}
void check(int rc, const char* what)
{
    if (rc != CAP_OK) error_throw(std::string(what) + ": " + cap_last_error());
}

struct AssetComponent
{
    std::string file_name;
    bool        loaded = false;
};

std::string dir_of(const std::string& p)
{
    size_t s = p.find_last_of('/');
    return s == std::string::npos ? std::string("") : p.substr(0, s + 1);
}

bool read_file(const std::string& path, std::vector<uint8_t>* out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    out->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}

// texture_system.cpp:41-45: stbi_load(file, &w, &h, &n, 4) -> own decoders behind the C ABI (JPEG, PNG, BMP, TGA, binary PNM)
bool decode_image(const std::vector<uint8_t>& d, const std::string& name, std::vector<uint8_t>* rgba, uint32_t* w, uint32_t* h)
{
    uint8_t* px = nullptr;
    if (cap_image_decode(d.data(), d.size(), name.c_str(), &px, w, h) != CAP_OK) return false;
    rgba->assign(px, px + (size_t)*w * *h * 4);
    cap_image_free(px);
    return true;
}

struct World
{
    // components
    std::vector<AssetComponent> assets;
    CameraData                  camera, prev_camera;  // CameraComponent::camera_data / prev_camera_data
    bool                        prev_camera_valid = false;
    Settings                    settings;
    // RenderSystem state
    RenderSessionParams session;
    bool                session_active = false;
    uint32_t            frame_count    = 0;
    // GPU side: one context per shard this process renders (ctx == ctxs[0]: the root of a single-process multi-GPU session)
    std::vector<CapContext*> ctxs;
    bool                     exchange = false;  // a gather + assembly ends every Render()
    CapContext* ctx        = nullptr;
    bool        tlas_built = false;  // tlas_system.cpp:111-121 `built` flag: the structure is built once
    // InputSystem state (input_system.h): the pending scripted input and the accumulated mouse angles, in degrees
    ScriptedInput input;
    bool          input_pending = false, angles_valid = false;
    float         yaw = 0.f, pitch = 0.f;
    float         forward_written[3] = {0.f, 0.f, 0.f};  // the forward axis run_input last wrote: a host that has since rewritten
                                                        // the basis through GetCamera() invalidates the accumulated angles
    std::string assets_dir;
};

std::unique_ptr<World> g_world;

World& world()
{
    if (!g_world) error_throw("capsaicin: Init() has not been called");
    return *g_world;
}

std::string find_assets_dir(const std::string& obj_file)
{
    if (const char* e = std::getenv("CAPSAICIN_ASSETS")) return std::string(e) + (e[std::strlen(e) - 1] == '/' ? "" : "/");
    return dir_of(obj_file);  // the reference hard-codes "../../../assets/" (asset_load_system.cpp:55)
}

// AssetLoadSystem::Run (asset_load_system.cpp:272-328) + TextureSystem (texture_system.cpp:38-118)
void run_asset_load(World& w)
{
    for (auto& a : w.assets)
    {
        if (a.loaded) continue;
        if (w.tlas_built) error_throw("AssetLoadSystem: the acceleration structure is built once; load scenes before the first Render()");
        info("AssetLoadSystem: loading " + a.file_name);
        w.assets_dir     = find_assets_dir(a.file_name);
        CapGeometry* geo = nullptr;
        check(cap_obj_load(a.file_name.c_str(), w.assets_dir.c_str(), &geo), "AssetLoadSystem");
        if (*cap_geometry_warning(geo)) warn(cap_geometry_warning(geo));
        CapGeometryView v;
        check(cap_geometry_view(geo, &v), "AssetLoadSystem");
        for (CapContext* c : w.ctxs) check(cap_scene_upload_geometry(c, geo), "AssetLoadSystem");  // the scene is replicated per GPU
        // Files are read and decoded on a few host threads, sixteen textures at a time (a scene's worth of 2048 x 2048 JPEGs is
        // seconds of decoding on one core); the uploads stay in texture order on this thread.
        struct Decoded
        {
            std::string          full;
            std::vector<uint8_t> rgba;
            uint32_t             w = 0, h = 0;
            bool                 ok = false;
        };
        constexpr uint32_t kBatch = 16;
        for (uint32_t first = 0; first < v.texture_count; first += kBatch)
        {
            const uint32_t       n = std::min(kBatch, v.texture_count - first);
            std::vector<Decoded> dec(n);
            for (uint32_t i = 0; i < n; ++i) dec[i].full = w.assets_dir + "textures/" + cap_geometry_texture_name(geo, first + i);
            std::atomic<uint32_t> next{0};
            auto                  work = [&]() {
                for (uint32_t i; (i = next.fetch_add(1)) < n;)
                {
                    try  // an exception must not leave a worker thread: the texture then counts as missing
                    {
                        std::vector<uint8_t> file;
                        dec[i].ok = read_file(dec[i].full, &file) && decode_image(file, dec[i].full, &dec[i].rgba, &dec[i].w, &dec[i].h);
                    }
                    catch (const std::exception&)
                    {
                        dec[i].ok = false;
                    }
                }
            };
            std::vector<std::thread> pool;
            const uint32_t           threads = std::min<uint32_t>(n, std::max(1u, std::thread::hardware_concurrency()));
            for (uint32_t k = 1; k < threads; ++k) pool.emplace_back(work);
            work();
            for (auto& th : pool) th.join();
            for (uint32_t i = 0; i < n; ++i)
            {
                const Decoded& d = dec[i];
                if (!d.ok) warn("TextureSystem: texture " + d.full + " missing or not decodable");  // texture_system.cpp:50-56
                for (CapContext* c : w.ctxs)
                    check(d.ok ? cap_texture_upload(c, first + i, d.rgba.data(), d.w, d.h) : cap_texture_upload(c, first + i, nullptr, 0, 0), "TextureSystem");
            }
        }
        cap_geometry_free(geo);
        a.loaded = true;
    }
}

// BLASSystem::Run + TLASSystem::Run (blas_system.cpp:69-113, tlas_system.cpp:81-122)
void run_acceleration_structure(World& w)
{
    if (w.tlas_built) return;
    bool any = false;
    for (auto& a : w.assets) any |= a.loaded;
    if (!any) return;
    for (CapContext* c : w.ctxs) check(cap_bvh_build(c), "TLASSystem");
    CapBvhInfo bi;
    check(cap_bvh_info(w.ctx, &bi), "TLASSystem");
    info("TLASSystem: LBVH over " + std::to_string(bi.triangle_count) + " triangles, depth " + std::to_string(bi.max_depth) + ", " +
         std::to_string(bi.build_ms) + " ms");
    w.tlas_built = true;
}

// CameraSystem::Run (camera_system.cpp:46-131) incl. AdjustCameraAspectBasedOnWindow (:10-17)
void run_camera(World& w)
{
    const float aspect       = float(w.session.height) / w.session.width;
    w.camera.sensor_size[1]  = w.camera.sensor_size[0] * aspect;
    CapCameraData cd;
    static_assert(sizeof(cd) == sizeof(CameraData), "layout");
    std::memcpy(&cd, &w.camera, sizeof(cd));
    for (CapContext* c : w.ctxs) check(cap_camera_set(c, &cd), "CameraSystem");
    // the first frame has no predecessor: prev = current (camera_system.cpp:104-118 uploads the stored previous data)
    if (!w.prev_camera_valid) w.prev_camera = w.camera, w.prev_camera_valid = true;
    w.prev_camera.sensor_size[1] = w.prev_camera.sensor_size[0] * aspect;
    std::memcpy(&cd, &w.prev_camera, sizeof(cd));
    for (CapContext* c : w.ctxs) check(cap_prev_camera_set(c, &cd), "CameraSystem");
}

// RaytracingSystem::Run, ray passes (raytracing_system.cpp:266-292)
void run_raytracing(World& w)
{
    if (!w.tlas_built) return;  // nothing to trace yet, as in the reference's first frames
    if (w.settings.reconstruct)
    {
        // ray passes + SpatialGather .. ApplyTAA of one frame (raytracing_system.cpp:262-317)
        const Settings& s = w.settings;
        if (w.exchange)  // the C ABI has the sharded chain (cap_post_frame_gathered, cap_feedback_export / _import); this layer does not drive it
            error_throw("RaytracingSystem: the reconstruction pipeline of this host layer renders on one GPU (gpus = 1, shard_count = 1)");
        if (s.frames_per_render != 1) error_throw("RaytracingSystem: the reconstruction pipeline renders one frame per Render()");
        const uint32_t flags = CAP_RENDER_STAGE_TIMERS | CAP_RENDER_AOV | (s.gbuffer_feedback ? (uint32_t)CAP_RENDER_GBUFFER_FEEDBACK : 0u) |
                               (s.lowres_indirect ? (uint32_t)CAP_RENDER_LOWRES_INDIRECT : 0u);
        check(cap_render(w.ctx, w.frame_count, 1, (uint32_t)std::max(0, s.num_diffuse_bounces), flags), "RaytracingSystem");
        CapPostSettings ps{s.gather, s.denoise, s.eaw5, s.eaw_normal_sigma, s.eaw_depth_sigma, s.eaw_luma_sigma, s.gather_normal_sigma,
                           s.gather_depth_sigma, s.gather_luma_sigma, s.temporal_upscale_feedback, s.taa_feedback, s.lowres_indirect,
                           s.use_variance ? 0 : 1, s.fast_weights, s.output};
        CapCameraData   prev;
        std::memcpy(&prev, &w.prev_camera, sizeof(prev));
        check(cap_post_frame(w.ctx, &ps, w.frame_count, &prev), "RaytracingSystem");
        return;
    }
    for (CapContext* c : w.ctxs)
    {
        if (!w.settings.accumulate) check(cap_accum_reset(c), "RaytracingSystem");
        check(cap_render(c, w.frame_count, w.settings.frames_per_render, (uint32_t)std::max(0, w.settings.num_diffuse_bounces),
                         CAP_RENDER_STAGE_TIMERS),
              "RaytracingSystem");
    }
    // frame end: the one collective of the design -- tile radiance to shard 0, assembled there
    if (w.exchange)
    {
        if (w.ctxs.size() > 1)
            check(cap_comm_gather_frame_all(w.ctxs.data(), (uint32_t)w.ctxs.size()), "RaytracingSystem");
        else
            check(cap_comm_gather_frame(w.ctx), "RaytracingSystem");
    }
}
}  // namespace

void Init()
{
    info("capsaicin::Init()");
    g_world.reset(new World);
}

void InitRenderSession(void* params)
{
    World& w = world();
    if (!params) error_throw("InitRenderSession: params is null");
    w.session = *static_cast<RenderSessionParams*>(params);
    w.angles_valid = false;  // a new session starts from the camera as it stands, not from an earlier session's mouse angles
    info("capsaicin::InitRenderSession()");
    const RenderSessionParams& sp = w.session;
    if (!sp.gpus || !sp.shard_count || sp.shard_index >= sp.shard_count) error_throw("InitRenderSession: bad shard description");
    if (sp.gpus > 1 && sp.shard_count > 1) error_throw("InitRenderSession: use gpus (one process, several GPUs) or shard_index / shard_count (one process per GPU), not both");
    // blue-noise texture = the sampler's random numbers (raytracing_system.cpp:642-646)
    std::vector<uint8_t> bn;
    std::string          dir = std::getenv("CAPSAICIN_ASSETS") ? std::string(std::getenv("CAPSAICIN_ASSETS")) + "/" : std::string("assets/");
    if (!read_file(dir + "bluenoise256.rgba", &bn) || bn.size() != 256 * 256 * 4)
        error_throw("RaytracingSystem: cannot read " + dir + "bluenoise256.rgba (set CAPSAICIN_ASSETS)");
    const int ndev = std::max(1, cap_device_count());
    for (uint32_t i = 0; i < sp.gpus; ++i)
    {
        CapContext* c = nullptr;
        check(cap_ctx_create((sp.device + (int)i) % ndev, nullptr, &c), "InitRenderSession");
        w.ctxs.push_back(c);
        check(cap_set_resolution(c, sp.width, sp.height), "InitRenderSession");
        check(sp.gpus > 1 ? cap_set_shard(c, i, sp.gpus) : cap_set_shard(c, sp.shard_index, sp.shard_count), "InitRenderSession");
        check(cap_bluenoise_upload(c, bn.data()), "RaytracingSystem");
    }
    w.ctx = w.ctxs[0];
    if (sp.gpus > 1)
    {
        check(cap_comm_init_all(w.ctxs.data(), sp.gpus), "InitRenderSession");
        w.exchange = true;
    }
    else if (sp.shard_count > 1)
    {
        if (!sp.comm_id) error_throw("InitRenderSession: shard_count > 1 needs comm_id (cap_comm_unique_id of rank 0)");
        check(cap_comm_init_rank(w.ctx, sp.comm_id, sp.shard_index, sp.shard_count), "InitRenderSession");
        w.exchange = true;
    }
    w.session_active = true;
}

void LoadSceneFromOBJ(const std::string& file_name)
{
    info("capsaicin::LoadSceneFromOBJ(" + file_name + ")");
    world().assets.push_back(AssetComponent{file_name, false});  // lazily loaded by the next Render(), capsaicin.cpp:65-71
}

void ProcessInput(void* input)
{
    World& w        = world();
    w.input_pending = input != nullptr;
    if (input) w.input = *static_cast<const ScriptedInput*>(input);
}
void Update(float) {}

namespace
{
// InputSystem::Run (input_system.cpp:13-34): HandleMouse, then HandleKeyboard, on the one camera.
void run_input(World& w)
{
    if (!w.input_pending) return;
    w.input_pending        = false;
    const ScriptedInput in = w.input;
    CameraData&         cd = w.camera;
    if (in.rotate)
    {
        // The reference's yaw_ / pitch_ start at 0 with the camera looking down +z (camera_system.cpp:25-33).  A session whose
        // camera was placed through GetCamera() continues from THAT view [not-ref]: the angles whose rotation gives its forward
        // vector, forward = (sin yaw cos pitch, -sin pitch, cos yaw cos pitch) -- see below.
        if (w.angles_valid && (cd.forward[0] != w.forward_written[0] || cd.forward[1] != w.forward_written[1] || cd.forward[2] != w.forward_written[2]))
            w.angles_valid = false;  // the host placed the camera itself since the last rotation: continue from its view
        if (!w.angles_valid)
        {
            const float fy = std::fmin(1.f, std::fmax(-1.f, cd.forward[1]));
            w.pitch        = -std::asin(fy) * (180.f / 3.14159265358979323846f);
            w.yaw          = std::atan2(cd.forward[0], cd.forward[2]) * (180.f / 3.14159265358979323846f);
            w.angles_valid = true;
        }
        w.yaw += in.dyaw_deg, w.pitch += in.dpitch_deg;   // input_system.cpp:122-123
        if (std::fabs(w.yaw) >= 360.f) w.yaw = 0.f;       // :125-128
        if (std::fabs(w.pitch) >= 360.f) w.pitch = 0.f;
        // XMMatrixRotationRollPitchYaw(pitch, yaw, 0) (:135-136) is, for DirectXMath's row vectors, Rx(pitch) * Ry(yaw) -- roll, then
        // pitch, then yaw -- with Rx = [1 0 0; 0 c s; 0 -s c] and Ry = [c 0 -s; 0 1 0; s 0 c].  (0, 0, 1) * Rx = (0, -sin p, cos p), and
        // that times Ry = (cos p sin y, -sin p, cos p cos y): XMVector3Transform of the forward axis (:138-139), then normalised.
        const float p = w.pitch * (3.14159265358979323846f / 180.f), y = w.yaw * (3.14159265358979323846f / 180.f);
        float       f[3] = {std::cos(p) * std::sin(y), -std::sin(p), std::cos(p) * std::cos(y)};
        const float fl   = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
        for (float& x : f) x /= fl;
        // right = normalize(-cross(forward, (0, 1, 0))), up = cross(forward, right)  (:140-146)
        float       r[3] = {-(f[1] * 0.f - f[2] * 1.f), -(f[2] * 0.f - f[0] * 0.f), -(f[0] * 1.f - f[1] * 0.f)};
        const float rl   = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        // Looking straight up or down (pitch = +-90 degrees) -cross(forward, (0, 1, 0)) has no length: the reference divides by it and
        // puts NaNs into the basis (input_system.cpp:140-143).  Here the previous right axis is kept [not-ref], so the basis stays
        // orthonormal and cap_camera_set never sees a NaN.
        if (rl > 0.f)
            for (float& x : r) x /= rl;
        else
            for (int k = 0; k < 3; ++k) r[k] = cd.right[k];
        const float u[3] = {f[1] * r[2] - f[2] * r[1], f[2] * r[0] - f[0] * r[2], f[0] * r[1] - f[1] * r[0]};
        for (int k = 0; k < 3; ++k) cd.forward[k] = f[k], cd.right[k] = r[k], cd.up[k] = u[k], w.forward_written[k] = f[k];
    }
    // HandleKeyboard (:50-108): the movement is summed along the (new) axes, then added to the position
    float movement[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; ++k)
    {
        movement[k] += cd.right[k] * in.move_right;
        movement[k] += cd.forward[k] * in.move_forward;
        movement[k] += cd.up[k] * in.move_up;
    }
    for (int k = 0; k < 3; ++k) cd.position[k] += movement[k];
}
}  // namespace
void SetOption() {}

void Render()
{
    World& w = world();
    if (!w.session_active) error_throw("Render: no render session");
    run_input(w);
    run_asset_load(w);
    run_acceleration_structure(w);
    run_camera(w);
    run_raytracing(w);
    // RenderSystem::Run: submit + ++frame_count_ (render_system.cpp:53-84)
    for (CapContext* c : w.ctxs) check(cap_sync(c), "RenderSystem");
    w.frame_count += w.settings.frames_per_render;
    w.prev_camera = w.camera;  // CameraSystem keeps this frame's data as the next frame's prev_camera_data
}

void ShutdownRenderSession()
{
    World& w = world();
    info("capsaicin::ShutdownRenderSession()");
    for (CapContext* c : w.ctxs) cap_ctx_destroy(c);
    w.ctxs.clear();
    w.ctx            = nullptr;
    w.session_active = false;
}

void Shutdown()
{
    info("capsaicin::Shutdown()");
    if (g_world)
        for (CapContext* c : g_world->ctxs) cap_ctx_destroy(c);
    g_world.reset();
}

Settings&   GetSettings() { return world().settings; }
CameraData& GetCamera() { return world().camera; }
uint32_t    FrameCount() { return world().frame_count; }

void ReadFrame(float* dst)
{
    World& w = world();
    if (w.settings.reconstruct)
        check(cap_post_readback(w.ctx, dst), "ReadFrame");  // current_frame_output(), raytracing_system.cpp:320-324
    else if (w.exchange)
        check(cap_comm_readback(w.ctx, dst), "ReadFrame");  // the frame shard 0 assembled from the gathered tiles
    else
        check(cap_readback(w.ctx, CAP_BUF_ACCUM_MEAN, dst), "ReadFrame");
}

void SaveFramePPM(const std::string& path)
{
    World&             w = world();
    std::vector<float> img((size_t)w.session.width * w.session.height * 4);
    ReadFrame(img.data());
    std::ofstream f(path, std::ios::binary);
    if (!f) error_throw("SaveFramePPM: cannot open " + path);
    f << "P6\n" << w.session.width << " " << w.session.height << "\n255\n";
    std::vector<uint8_t> row(3 * (size_t)w.session.width);
    for (uint32_t y = 0; y < w.session.height; ++y)
    {
        const float* src = img.data() + 4 * (size_t)(w.session.height - 1 - y) * w.session.width;  // simple.hlsl:44 flips v
        for (uint32_t x = 0; x < w.session.width; ++x)
            for (int c = 0; c < 3; ++c)
            {
                float v = std::pow(std::fmax(src[4 * x + c], 0.f), 1.f / 2.2f);  // simple.hlsl:45
                row[3 * x + c] = (uint8_t)std::fmin(255.f, std::floor(v * 255.f + 0.5f));
            }
        f.write((const char*)row.data(), row.size());
    }
}

std::string TimingsReport()
{
    CapStats s;
    check(cap_stats_get(world().ctx, &s), "TimingsReport");
    for (size_t i = 1; i < world().ctxs.size(); ++i)
    {
        // the other shards of this process: their rays add up, their passes run concurrently on their own GPUs
        CapStats o;
        check(cap_stats_get(world().ctxs[i], &o), "TimingsReport");
        s.rays_primary += o.rays_primary, s.rays_extension += o.rays_extension, s.rays_shadow += o.rays_shadow;
    }
    // the reference's timestamp labels (gui_system.cpp:94-104 prints what the passes named with AllocateTimestampQueryPair:
    // raytracing_system.cpp:1024, 1099, 1207 and the reconstruction passes).  The wavefront passes map onto them as follows:
    // camera rays -> "RaytracePrimaryVisibility" (the fused small-scene kernel also shades the camera vertex there), the camera
    // vertex's shading and shadow rays -> "RT Direct lighting", everything of bounce >= 1 -> "RT Indirect diffuse".
    const double indirect = s.ms_shade + s.ms_trace_closest + s.ms_trace_any - s.ms_direct;
    std::ostringstream o;
    o << "RaytracePrimaryVisibility: " << s.ms_primary << " ms\n"
      << "RT Direct lighting: " << s.ms_direct << " ms\n"
      << "RT Indirect diffuse: " << indirect << " ms\n"
      << "Spatial gather: " << s.ms_post_pass[0] << " ms\n"
      << "Temporal upscale: " << s.ms_post_pass[1] << " ms\n"
      << "EAW: " << s.ms_post_pass[2] << " ms\n"
      << "Combine illumination: " << s.ms_post_pass[3] << " ms\n"
      << "TAA: " << s.ms_post_pass[4] << " ms\n"
      << "Accumulate: " << s.ms_resolve << " ms\n"
      << "total: " << s.ms_total << " ms ray passes over " << s.frames << " frames, " << s.ms_post << " ms reconstruction over "
      << s.post_frames << " frames, rays primary/extension/shadow = " << s.rays_primary << "/" << s.rays_extension << "/" << s.rays_shadow
      << "\n";
    return o.str();
}
}  // namespace capsaicin
