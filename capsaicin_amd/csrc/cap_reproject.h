// cap_reproject.h — device helpers shared by the reconstruction chain (post.hip) and the G-buffer feedback branch of the
// indirect pass (kernels.hip): row-major float4 images with D3D-style out-of-bounds reads, the reference's UV<->pixel rules,
// its hand-written bilinear tap and the camera reprojection (reference utils.h:6-35, camera.h:8-37).
#pragma once

#include "cap_device.h"

namespace cap
{
struct Img
{
    const float4* p;
    uint32_t      w, h;
};
struct f2
{
    float x, y;
};

__device__ __forceinline__ float4 ld(const Img& t, uint32_t x, uint32_t y)  // out-of-bounds reads return 0 (D3D UAV rule)
{
    return (x < t.w && y < t.h) ? t.p[(size_t)y * t.w + x] : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float4 ldi(const Img& t, int x, int y)
{
    return (x >= 0 && y >= 0) ? ld(t, (uint32_t)x, (uint32_t)y) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ v3       xyz(float4 v) { return mk3(v.x, v.y, v.z); }
__device__ __forceinline__ uint32_t sat_uint(float f) { return f > 0.0f ? (uint32_t)f : 0u; }  // uint(x): negative saturates to 0
__device__ __forceinline__ float    frac1(float x) { return x - floorf(x); }
__device__ __forceinline__ v3       lerp3(v3 a, v3 b, float t) { return mk3(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)); }

// utils.h:6-16
__device__ __forceinline__ f2 uv_to_xy(f2 uv, uint32_t w, uint32_t h)
{
    return f2{fminf(uv.x * (float)w, (float)(w - 1)), fminf(uv.y * (float)h, (float)(h - 1))};
}
__device__ __forceinline__ f2 xy_to_uv(f2 xy, uint32_t w, uint32_t h)
{
    return f2{fminf(fmaxf(xy.x / (float)w, 0.0f), 1.0f), fminf(fmaxf(xy.y / (float)h, 0.0f), 1.0f)};
}
// utils.h:20-35
__device__ __forceinline__ v3 sample_bilinear(const Img& t, f2 uv)
{
    const f2       xy = uv_to_xy(uv, t.w, t.h);
    const float    fx = xy.x - 0.5f, fy = xy.y - 0.5f;
    const uint32_t ux = sat_uint(floorf(fx)), uy = sat_uint(floorf(fy));
    const float    wx = frac1(fx), wy = frac1(fy);
    const v3 v00 = xyz(ld(t, ux, uy)), v01 = xyz(ld(t, ux, uy + 1)), v10 = xyz(ld(t, ux + 1, uy)), v11 = xyz(ld(t, ux + 1, uy + 1));
    return lerp3(lerp3(v00, v10, wx), lerp3(v01, v11, wx), wy);
}

// camera.h:8-37 CalculateImagePlaneUV
__device__ __forceinline__ f2 image_plane_uv(const CameraDev& cam, v3 position)
{
    const v3    o = mk3(cam.position[0], cam.position[1], cam.position[2]);
    const v3    d = normalize3(position - o);
    const v3    n = normalize3(mk3(cam.forward[0], cam.forward[1], cam.forward[2]));
    const v3    p = o + n * cam.focal_length;
    const float t = dot3(n, p - o) / dot3(n, d);
    const v3    ip = o + d * t;
    const v3    ipd = ip - p;
    const float u = dot3(mk3(cam.right[0], cam.right[1], cam.right[2]), ipd) / (0.5f * cam.sensor_x);
    const float v = dot3(mk3(cam.up[0], cam.up[1], cam.up[2]), ipd) / (0.5f * cam.sensor_y);
    return f2{0.5f * u + 0.5f, 0.5f * v + 0.5f};
}
}  // namespace cap
