// capsaicin.h — host API with the reference's names and argument meaning (reference src/core/include/capsaicin.h:25-36),
// for a display-less MI355X node.  The nine entry points keep their semantics; what was a Win32 HWND in
// RenderSessionParams is a plain headless description, and the frame is fetched with the accessors below instead
// of a swap chain (CompositeSystem blit, simple.hlsl:40-46).
#pragma once

#include <cstdint>
#include <string>

struct RenderSessionParams
{
    uint32_t width       = 1920;  // viewer window size, main.cpp:53-54
    uint32_t height      = 1080;
    int      device      = 0;     // HIP device
    // Multi-GPU (no reference counterpart: dx12.cpp:13-25 picks one adapter).  Frames shard by 8x8 screen tile, tile t -> shard
    // t % count; one RCCL gather of tile radiance to shard 0 per Render(), which assembles the frame (cap_comm_*).
    //  * one process driving several GPUs: gpus = N; shard i renders on HIP device (device + i) % device count.  Shards that end
    //    up on one device exchange by device copies.
    //  * one process per GPU: shard_index / shard_count = this process's rank / the number of ranks, comm_id = the 128 bytes of
    //    cap_comm_unique_id() made by rank 0 and carried to every rank by the launcher; ReadFrame / SaveFramePPM on rank 0.
    uint32_t       gpus        = 1;
    uint32_t       shard_index = 0;
    uint32_t       shard_count = 1;
    const uint8_t* comm_id     = nullptr;
};

// What ProcessInput() takes on a display-less node instead of the viewer's Win32 message (main.cpp:12-16 -> InputSystem::ProcessInput,
// input_system.cpp:36-48): one frame's worth of what the keyboard and the mouse would have added up to.  Consumed by the next
// Render(), whose InputSystem step applies it the way the reference does (input_system.cpp:31-33: mouse first, then keyboard).
struct ScriptedInput
{
    // HandleKeyboard (input_system.cpp:50-108): displacement along the camera's right (D - A), up (E - Q) and forward (W - S) axes,
    // i.e. kMovementSpeed * dt per key held
    float move_right = 0.f, move_up = 0.f, move_forward = 0.f;
    // HandleMouse (input_system.cpp:109-147): what (mouse delta) * kMouseSensitivity * dt adds to yaw_ / pitch_, in degrees; the
    // camera basis is rebuilt from (pitch_, yaw_) only when the button is down, i.e. when `rotate` is set
    float dyaw_deg = 0.f, dpitch_deg = 0.f;
    bool  rotate = false;
};

namespace capsaicin
{
// Settings carried from SettingsComponent (gui_system.h:20-40); only what the ray passes read.
struct Settings
{
    int      num_diffuse_bounces = 1;   // gui_system.h:39
    uint32_t frames_per_render   = 1;   // reference: exactly one frame per Render()
    bool     accumulate          = true;  // plain running mean instead of the temporal EMA
    // The reference's own frame pipeline (raytracing_system.cpp:262-317): one frame per Render(), reconstruction chain after
    // the ray passes, G-buffer feedback in the indirect pass.  ReadFrame() then returns current_frame_output().
    bool  reconstruct               = false;
    bool  gbuffer_feedback          = true;    // RaytracingOptions::gbuffer_feedback, raytracing_system.h:26
    bool  lowres_indirect           = false;   // RaytracingOptions::lowres_indirect, raytracing_system.h:24 (even window sizes)
    bool  use_variance              = true;    // RaytracingOptions::use_variance, raytracing_system.h:25
    bool  fast_weights              = false;   // not a reference option: CapPostSettings::fast_weights (toleranced chain)
    int   output                    = 0;       // SettingsComponent::output, gui_system.h:11-17, 38: 0 kCombined, 1 kDirect, 2 kIndirect, 3 kVariance
    bool  gather                    = true;    // gui_system.h:20-37
    bool  denoise                   = true;
    bool  eaw5                      = true;
    float eaw_normal_sigma          = 128.f;
    float eaw_depth_sigma           = 3.f;
    float eaw_luma_sigma            = 3.f;
    float gather_normal_sigma       = 64.f;
    float gather_depth_sigma        = 2.f;
    float gather_luma_sigma         = 3.f;
    float temporal_upscale_feedback = 0.975f;
    float taa_feedback              = 0.9f;
};

// CameraData (camera_system.h:16-31), defaults from CameraSystem's ctor (camera_system.cpp:25-33).
struct CameraData
{
    float position[3]    = {0.f, 15.f, 0.f};
    float focal_length   = 0.016f;
    float right[3]       = {1.f, 0.f, 0.f};
    float znear          = 0.f;
    float forward[3]     = {0.f, 0.f, 1.f};
    float focus_distance = 0.f;
    float up[3]          = {0.f, 1.f, 0.f};
    float aperture       = 0.f;
    float sensor_size[2] = {0.036f, 0.024f};
};
static_assert(sizeof(CameraData) == 72, "CameraData must stay the reference's 72-byte POD");

void Init();
void InitRenderSession(void* params);  // RenderSessionParams*
void LoadSceneFromOBJ(const std::string& file_name);
void ProcessInput(void* input);  // ScriptedInput* (or nullptr: nothing happened); applied by the next Render()
void Update(float time_ms);
void Render();
void SetOption();  // empty stub in the reference as well (capsaicin.cpp:89-92)
void ShutdownRenderSession();
void Shutdown();

// ---- headless additions ----
Settings&   GetSettings();
CameraData& GetCamera();          // edit, then Render(); the sensor height follows the window aspect (camera_system.cpp:10-17)
uint32_t    FrameCount();         // RenderSystem::frame_count()
// Linear radiance (running mean) of the current image, width*height*4 floats, row 0 = pixel row 0.
void ReadFrame(float* dst_rgba);
// Gamma 1/2.2 8-bit PPM with the vertical flip of the reference blit (simple.hlsl:40-46).
void SaveFramePPM(const std::string& path);
// Per-pass GPU milliseconds under the reference's timestamp names (gui_system.cpp:94-104).
std::string TimingsReport();
}  // namespace capsaicin
