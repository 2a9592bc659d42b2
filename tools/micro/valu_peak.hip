// Microbenchmark: sustained wave64 v_fma_f32 issue rate per SIMD on gfx950, as a function of waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_peak.hip -o gpurun_out/valu_peak && gpurun_out/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float a, float b)
{
    float x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 0.001f + c;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fmaf(x[c], a, b);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    if (s == 12345.678f) out[0] = s;
}

template <int CHAINS>
void run(int blocks_per_cu, int cus, float* d)
{
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k_fma<CHAINS><<<cus * blocks_per_cu, 256>>>(d, 16, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_fma<CHAINS><<<cus * blocks_per_cu, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)blocks_per_cu * iters * 16 * CHAINS;  // one wave of each block per SIMD
    // report cycles per instruction per SIMD at a nominal 2.4 GHz
    printf("chains %d waves/SIMD %d: %.3f ms, %.2f nominal cycles per wave64 fma per SIMD\n", CHAINS, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    float* d;
    hipMalloc(&d, 4);
    for (int w : {1, 2, 4, 8})
    {
        run<1>(w, p.multiProcessorCount, d);
        run<4>(w, p.multiProcessorCount, d);
    }
    return 0;
}
