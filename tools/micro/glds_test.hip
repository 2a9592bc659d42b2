// Checks the semantics of __builtin_amdgcn_global_load_lds (16-byte form) on gfx950 before relying on it: per-lane global
// source, LDS destination = wave-uniform base + lane * 16.
// hipcc --offload-arch=gfx950 -O3 tools/micro/glds_test.hip -o gpurun_out/glds_test && gpurun_out/glds_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k(const float4* src, const uint32_t* perm, float4* out)
{
    __shared__ float4 buf[256];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t i = perm[blockIdx.x * 256 + threadIdx.x];
    float4* base = buf + wave * 64;  // wave-uniform
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i),
                                     (__attribute__((address_space(3))) void*)base, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the DMA has landed
    __builtin_amdgcn_wave_barrier();
    out[blockIdx.x * 256 + threadIdx.x] = buf[wave * 64 + lane];
}

int main()
{
    const int n = 256 * 64;
    std::vector<float4> h(n);
    std::vector<uint32_t> p(n);
    for (int i = 0; i < n; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f), p[i] = (uint32_t)((i * 7919u + 13u) % n);
    float4 *d, *o;
    uint32_t* dp;
    hipMalloc(&d, n * 16), hipMalloc(&o, n * 16), hipMalloc(&dp, n * 4);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice), hipMemcpy(dp, p.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, dp, o);
    std::vector<float4> r(n);
    if (hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += r[i].x != h[p[i]].x || r[i].w != h[p[i]].w;
    printf("glds 16-byte per-lane source -> lane-linear LDS: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
    return bad != 0;
}
