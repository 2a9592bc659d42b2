// viewer_main.cpp — headless counterpart of the reference viewer (src/viewer/main.cpp:50-107): same call sequence
// (Init, InitRenderSession, LoadSceneFromOBJ, Render per frame, ShutdownRenderSession, Shutdown), a frame loop
// instead of the Win32 message pump, a PPM file instead of the swap chain.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "capsaicin.h"

int main(int argc, char** argv)
{
    RenderSessionParams params;  // 1920x1080 like main.cpp:53-54
    std::string         scene = "assets/cornell_box.obj", out = "frame.ppm";
    int                 frames = 64, bounces = 1;
    bool                cornell_camera = true, realtime = false, feedback = true, lowres = false;
    float               move[3] = {0.f, 0.f, 0.f};  // camera translation per frame (a scripted fly-through, input_system.cpp:49-148)
    float               view[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // --camera: position, forward, focal length
    bool                custom_view = false, print_cameras = false;
    std::string         script;  // --script: one line per frame, "right up forward dyaw dpitch" (ScriptedInput; degrees)
    for (int i = 1; i < argc; ++i)
    {
        auto next = [&]() { return i + 1 < argc ? argv[++i] : ""; };
        if (!std::strcmp(argv[i], "--scene")) scene = next();
        else if (!std::strcmp(argv[i], "--out")) out = next();
        else if (!std::strcmp(argv[i], "--width")) params.width = (uint32_t)std::atoi(next());
        else if (!std::strcmp(argv[i], "--height")) params.height = (uint32_t)std::atoi(next());
        else if (!std::strcmp(argv[i], "--frames")) frames = std::atoi(next());
        else if (!std::strcmp(argv[i], "--bounces")) bounces = std::atoi(next());
        else if (!std::strcmp(argv[i], "--device")) params.device = std::atoi(next());
        else if (!std::strcmp(argv[i], "--gpus")) params.gpus = (uint32_t)std::atoi(next());  // screen-tile shards, one per GPU
        else if (!std::strcmp(argv[i], "--default-camera")) cornell_camera = false;
        else if (!std::strcmp(argv[i], "--realtime")) realtime = true;  // the reference pipeline: 1 spp per frame + reconstruction chain
        else if (!std::strcmp(argv[i], "--no-feedback")) feedback = false;
        else if (!std::strcmp(argv[i], "--lowres")) lowres = true;  // half-resolution interleaved indirect (lowres_indirect)
        else if (!std::strcmp(argv[i], "--move"))
            for (int k = 0; k < 3; ++k) move[k] = (float)std::atof(next());
        else if (!std::strcmp(argv[i], "--script")) script = next();
        else if (!std::strcmp(argv[i], "--print-cameras")) print_cameras = true;  // the camera of every frame on stderr, as hex floats
        else if (!std::strcmp(argv[i], "--camera"))
        {
            for (int k = 0; k < 7; ++k) view[k] = (float)std::atof(next());
            custom_view = true;
        }
        else
        {
            std::fprintf(stderr, "usage: %s [--scene f.obj] [--out f.ppm] [--width W] [--height H] [--frames N] [--bounces D] [--device i] [--gpus N] [--default-camera | --camera px py pz fx fy fz focal] [--realtime [--no-feedback] [--lowres] [--move dx dy dz] [--script file] [--print-cameras]]\n", argv[0]);
            return 2;
        }
    }
    try
    {
        capsaicin::Init();
        capsaicin::InitRenderSession(&params);
        capsaicin::LoadSceneFromOBJ(scene);
        capsaicin::GetSettings().num_diffuse_bounces = bounces;
        capsaicin::GetSettings().reconstruct         = realtime;
        capsaicin::GetSettings().gbuffer_feedback    = feedback;
        capsaicin::GetSettings().lowres_indirect     = lowres;
        if (custom_view)
        {
            // right and up as InputSystem derives them from the forward vector (input_system.cpp:134-141)
            auto&       cam = capsaicin::GetCamera();
            const float fl  = std::sqrt(view[3] * view[3] + view[4] * view[4] + view[5] * view[5]);
            const float f[3] = {view[3] / fl, view[4] / fl, view[5] / fl};
            float       r[3] = {-(f[1] * 0.f - f[2] * 1.f), -(f[2] * 0.f - f[0] * 0.f), -(f[0] * 1.f - f[1] * 0.f)};  // -cross(f, (0,1,0))
            const float rl   = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            for (float& x : r) x /= rl;
            const float u[3] = {f[1] * r[2] - f[2] * r[1], f[2] * r[0] - f[0] * r[2], f[0] * r[1] - f[1] * r[0]};  // cross(f, right)
            for (int k = 0; k < 3; ++k) cam.position[k] = view[k], cam.forward[k] = f[k], cam.right[k] = r[k], cam.up[k] = u[k];
            cam.focal_length = view[6];
        }
        else if (cornell_camera)
        {
            // the reference default (0,15,0)/+z is tuned for Sponza; SURVEY.md 8d fixes this view for the Cornell box
            auto& cam = capsaicin::GetCamera();
            cam.position[0] = -0.01f, cam.position[1] = 0.995f, cam.position[2] = 3.4f;
            cam.forward[0] = 0.f, cam.forward[1] = 0.f, cam.forward[2] = -1.f;
            cam.right[0] = -1.f, cam.right[1] = 0.f, cam.right[2] = 0.f;
            cam.up[0] = 0.f, cam.up[1] = 1.f, cam.up[2] = 0.f;
            cam.focal_length = 0.035f;
        }
        // fly-camera script (the replay of what InputSystem would have seen, input_system.cpp:49-148): line f is applied by frame f's
        // Render(); a frame without a line gets no input
        std::vector<ScriptedInput> steps;
        if (!script.empty())
        {
            std::FILE* sf = std::fopen(script.c_str(), "r");
            if (!sf)
            {
                std::fprintf(stderr, "fatal: cannot open script %s\n", script.c_str());
                return 1;
            }
            char line[256];
            while (std::fgets(line, sizeof(line), sf))
            {
                if (line[0] == '#' || line[0] == '\n') continue;
                ScriptedInput in;
                if (std::sscanf(line, "%f %f %f %f %f", &in.move_right, &in.move_up, &in.move_forward, &in.dyaw_deg, &in.dpitch_deg) != 5)
                {
                    std::fprintf(stderr, "fatal: script line is not `right up forward dyaw dpitch`: %s", line);
                    std::fclose(sf);
                    return 1;
                }
                in.rotate = in.dyaw_deg != 0.f || in.dpitch_deg != 0.f;  // the mouse button is down exactly while it turns the view
                steps.push_back(in);
            }
            std::fclose(sf);
        }
        for (int f = 0; f < frames; ++f)
        {
            if ((size_t)f < steps.size()) capsaicin::ProcessInput(&steps[(size_t)f]);
            capsaicin::Render();  // one Render() per WM_PAINT in the reference (main.cpp:17-19)
            if (print_cameras)
            {
                const auto& c = capsaicin::GetCamera();
                std::fprintf(stderr, "camera %d: %a %a %a  %a %a %a  %a %a %a  %a %a %a\n", f, c.position[0], c.position[1], c.position[2], c.forward[0],
                             c.forward[1], c.forward[2], c.right[0], c.right[1], c.right[2], c.up[0], c.up[1], c.up[2]);
            }
            for (int k = 0; k < 3; ++k) capsaicin::GetCamera().position[k] += move[k];
        }
        capsaicin::SaveFramePPM(out);
        std::fputs(capsaicin::TimingsReport().c_str(), stderr);
        capsaicin::ShutdownRenderSession();
        capsaicin::Shutdown();
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "fatal: %s\n", e.what());
        return 1;
    }
    return 0;
}
