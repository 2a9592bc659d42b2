// cap_trace.h — device helpers shared by the traversal kernels (kernels.hip, trace8.hip): the ray record, the ray / triangle
// tests of the intersection contract and the chunk distribution of the persistent queue kernels.
#pragma once

#include "cap_device.h"

namespace cap
{
// ------------------------------------------------------------------------------------------------
// Traversal
// ------------------------------------------------------------------------------------------------
struct Ray
{
    v3    o, d, inv;
    float tmin, tmax;
};

__device__ __forceinline__ Ray make_ray(v3 o, v3 d, float tmin, float tmax)
{
    Ray r;
    r.o = o, r.d = d, r.tmin = tmin, r.tmax = tmax;
    r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    return r;
}

// Conservative slab test: returns entry distance, hit when entry <= exit * (1 + 2ulp).  NaNs from 0 * inf are
// dropped by fminf/fmaxf (IEEE minNum/maxNum), which only widens the interval.
__device__ __forceinline__ bool slab(const Ray& r, float lox, float loy, float loz, float hix, float hiy, float hiz, float tfar,
                                     float& tnear_out)
{
    const float ax = (lox - r.o.x) * r.inv.x, bx = (hix - r.o.x) * r.inv.x;
    const float ay = (loy - r.o.y) * r.inv.y, by = (hiy - r.o.y) * r.inv.y;
    const float az = (loz - r.o.z) * r.inv.z, bz = (hiz - r.o.z) * r.inv.z;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), r.tmin));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tfar));
    tnear_out      = tn;
    return tn <= tf * 1.0000004f;
}

// Ray / triangle test of the intersection contract: two-sided Moller-Trumbore in the determinant-scaled
// domain, regrouped around the per-triangle plane normal n = e1 x e2 and q = tvec x d:
//   det = e1.(d x e2) = -d.n    U = tvec.(d x e2) = e2.q    V = d.(tvec x e1) = -e1.q    T = e2.(tvec x e1) = tvec.n
// Hit rule: det != 0, U, V >= 0, U + V <= |det|, tmin < T/det < tmax (DXR triangle rule); barycentrics weight v1, v2
// (scene.h:46-49).  Written without branches: on this machine the scalar unit is shared by the four SIMDs of a CU, and the
// exec-mask bookkeeping of a branchy test costs more scalar issue slots than the vector work it skips.
__device__ __forceinline__ bool tri_test(const Ray& r, const float4 t0, const float4 t1, const float4 t2, float& t, float& u, float& v)
{
    const v3 v0 = mk3(t0.x, t0.y, t0.z), e1 = mk3(t0.w, t1.x, t1.y), e2 = mk3(t1.z, t1.w, t2.x), n = mk3(t2.y, t2.z, t2.w);
    const v3 tvec = r.o - v0;
    const v3 q    = cross3(tvec, r.d);
    float    det  = -dot3(r.d, n);
    float    U = dot3(e2, q), V = -dot3(e1, q), T = dot3(tvec, n);
    // two-sided: when det < 0 every sign flips (exact, so a sign-bit xor)
    const uint32_t sgn = f2u(det) & 0x80000000u;
    det = u2f(f2u(det) ^ sgn), U = u2f(f2u(U) ^ sgn), V = u2f(f2u(V) ^ sgn), T = u2f(f2u(T) ^ sgn);
    // det == 0 needs no test of its own: then inv overflows and tt is +-inf or NaN, which the interval test rejects
    // (det NaN fails every comparison), exactly where the contract's explicit det > 0 rejects.
    const bool  inside = (U >= 0.0f) & (V >= 0.0f) & (U + V <= det);
    const float inv    = rcp_c(det);
    const float tt     = T * inv;
    t = tt, u = U * inv, v = V * inv;
    return inside & (tt > r.tmin) & (tt < r.tmax);
}

// Occlusion form of the same test (any-hit queries observe only "is there a hit"): the open interval is checked in the scaled
// domain, tmin*det < T < tmax*det, so the shadow-ray kernels need no division at all.
__device__ __forceinline__ bool tri_occludes(const Ray& r, const float4 t0, const float4 t1, const float4 t2)
{
    const v3 v0 = mk3(t0.x, t0.y, t0.z), e1 = mk3(t0.w, t1.x, t1.y), e2 = mk3(t1.z, t1.w, t2.x), n = mk3(t2.y, t2.z, t2.w);
    const v3 tvec = r.o - v0;
    const v3 q    = cross3(tvec, r.d);
    float    det  = -dot3(r.d, n);
    float    U = dot3(e2, q), V = -dot3(e1, q), T = dot3(tvec, n);
    const uint32_t sgn = f2u(det) & 0x80000000u;
    det = u2f(f2u(det) ^ sgn), U = u2f(f2u(U) ^ sgn), V = u2f(f2u(V) ^ sgn), T = u2f(f2u(T) ^ sgn);
    return (det > 0.0f) & (U >= 0.0f) & (V >= 0.0f) & (U + V <= det) & (T > r.tmin * det) & (T < r.tmax * det);
}

// Work distribution of the queue kernels: the grid is persistent (fixed size, independent of the device-side
// queue length); each wave takes 64-ray chunks strided by the number of waves in the grid.  A wave whose first
// chunk is past the end leaves at once, so the grid always drains.
__device__ __forceinline__ uint32_t wave_global_id() { return blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); }
__device__ __forceinline__ uint32_t wave_total() { return gridDim.x * (kBlock / 64); }
// Cross-lane hand-off inside ONE wave through LDS or global memory (per-wave rings and ray buffers): what the writers stored
// before it is visible to the readers after it.  Wavefront-scope release + acquire (no instructions on gfx950: a wave's LDS and
// vector-memory operations retire in issue order) plus a scheduling barrier; the fences are what binds the compiler.
__device__ __forceinline__ void wave_handoff()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Dynamic chunk distribution.  A wave works on the chunk slots of ONE class (its index mod kQueueClasses) and takes them in order
// from that class's work counter (64 counters on their own 128-B lines, zeroed per launch; <= 5 grabs per microsecond each).
// Measured need (tools/wave_times.py): with a static slot-to-wave assignment the CU's oldest waves win the issue arbitration,
// finish their share at half the kernel's duration and leave the tail to a few starved young waves (mean wave life 70 % of the
// launch).  The grab for the NEXT chunk is issued before the current chunk is processed, so its latency is never waited for.
__device__ __forceinline__ uint32_t grab_issue(uint32_t* work, uint32_t klass)
{
    uint32_t v = 0;
    if ((threadIdx.x & 63u) == 0) v = atomicAdd(work + klass * kCounterStride, 1u);
    return v;  // lane 0 holds the value; grab_value() broadcasts it
}
__device__ __forceinline__ uint32_t grab_value(uint32_t raw) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)raw); }

}  // namespace cap
