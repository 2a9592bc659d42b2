// half_exec.hip -- does a wave64 vector instruction cost less when only one 32-lane half of EXEC is set?  (CDNA4 executes a wave64
// instruction as two 32-lane passes; if an all-zero half is skipped, a kernel can run two independent 32-lane groups per wave --
// each with its own loads in flight and its own s_waitcnt point -- at no extra issue cost.)
// hipcc --offload-arch=gfx950 -O3 tools/micro/half_exec.hip -o gpurun_out/half_exec && gpurun_out/half_exec
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>  // 0: all 64 lanes; 1: lanes 0..31; 2: lanes 32..63; 3: even lanes; 4: lanes 0..15; 5: lane 0 only
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b)
{
    const unsigned lane = threadIdx.x & 63u;
    const bool on = MODE == 0 ? true : MODE == 1 ? lane < 32u : MODE == 2 ? lane >= 32u : MODE == 3 ? (lane & 1u) == 0u : MODE == 4 ? lane < 16u : lane == 0u;
    float x0 = threadIdx.x * 0.001f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    if (on)
    {
        for (int i = 0; i < iters; ++i)
        {
#pragma unroll
            for (int u = 0; u < 16; ++u)
            {
                x0 = __builtin_fmaf(x0, a, b), x1 = __builtin_fmaf(x1, a, b), x2 = __builtin_fmaf(x2, a, b), x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b), x5 = __builtin_fmaf(x5, a, b), x6 = __builtin_fmaf(x6, a, b), x7 = __builtin_fmaf(x7, a, b);
            }
        }
    }
    const float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int per_cu, int cus, float* d)
{
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<MODE><<<cus * per_cu, 256>>>(d, 16, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<cus * per_cu, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)per_cu * iters * 16 * 8;  // wave instructions per SIMD
    printf("%-22s %d waves/SIMD  %8.3f ms  %.2f ns per wave-instruction per SIMD\n", name, per_cu, ms, ms * 1e6 / inst);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    float* d;
    hipMalloc(&d, 4096);
    for (int per_cu : {1, 4})
    {
        run<0>("all 64 lanes", per_cu, p.multiProcessorCount, d);
        run<1>("lanes 0..31", per_cu, p.multiProcessorCount, d);
        run<2>("lanes 32..63", per_cu, p.multiProcessorCount, d);
        run<3>("even lanes", per_cu, p.multiProcessorCount, d);
        run<4>("lanes 0..15", per_cu, p.multiProcessorCount, d);
        run<5>("lane 0", per_cu, p.multiProcessorCount, d);
    }
    return 0;
}
