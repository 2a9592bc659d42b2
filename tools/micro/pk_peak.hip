// Microbenchmark: wave64 v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 issue cost per SIMD on gfx950 next to v_fma_f32.
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/pk_peak.hip -o /tmp/pk_peak && /tmp/pk_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int CHAINS, int OP>
__global__ __launch_bounds__(256) void k_pk(float* out, int iters, float a, float b)
{
    const long long c0 = clock64(), w0 = wall_clock64();
    v2f x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = v2f{threadIdx.x * 0.001f + c, threadIdx.x * 0.002f - c};
    const v2f a2 = {a, a + 0.0001f}, b2 = {b, b - 0.25f};
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c)
            {
                if (OP == 0) x[c] = __builtin_elementwise_fma(x[c], a2, b2);
                if (OP == 1) x[c] = x[c] * a2;
                if (OP == 2) x[c] = x[c] + b2;
                if (OP == 3) x[c].x = __builtin_fmaf(x[c].x, a, b);  // scalar reference
                if (OP == 4) x[c].x = __builtin_fmaf(x[c].y, a, x[c].x);  // v_fmac_f32_e32 (VOP2)
                if (OP == 5) x[c].x = x[c].x * a;                         // v_mul_f32_e32 (VOP2)
                if (OP == 6) x[c].x = x[c].x + b2.y;                      // v_add_f32_e32 (VOP2)
                if (OP == 7) x[c].x = __builtin_fmaf(x[c].x, x[c].y, b2.y);  // v_fma_f32, VGPR sources only
                if (OP == 8) x[c].x = __int_as_float(__float_as_int(x[c].x) ^ __float_as_int(x[c].y));  // v_xor_b32_e32
            }
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += x[c].x + x[c].y;
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        // shader cycles (s_memtime) and 100-MHz wall ticks of this wave: the clock the SIMD actually ran at
        ((long long*)out)[1] = clock64() - c0;
        ((long long*)out)[2] = wall_clock64() - w0;
    }
}

template <int CHAINS, int OP>
void run(const char* name, int blocks_per_cu, int cus, float* d)
{
    const int  iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k_pk<CHAINS, OP><<<cus * blocks_per_cu, 256>>>(d, 16, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_pk<CHAINS, OP><<<cus * blocks_per_cu, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)blocks_per_cu * iters * 16 * CHAINS;
    long long t[3];
    hipMemcpy(t, d, sizeof(t), hipMemcpyDeviceToHost);
    const double ghz = (double)t[1] / ((double)t[2] * 10.0);  // cycles per ns
    printf("%-12s chains %2d waves/SIMD %d: %.3f ms, clock %.2f GHz, %.2f cycles per wave64 instruction per SIMD\n", name, CHAINS,
           blocks_per_cu, ms, ghz, ms * 1e-3 * ghz * 1e9 / instr_per_simd);
}

template <int CHAINS>
void row(int w, int cus, float* d)
{
    run<CHAINS, 3>("v_fma_f32", w, cus, d);
    run<CHAINS, 0>("v_pk_fma_f32", w, cus, d);
    run<CHAINS, 1>("v_pk_mul_f32", w, cus, d);
    run<CHAINS, 2>("v_pk_add_f32", w, cus, d);
    run<CHAINS, 4>("v_fmac_e32", w, cus, d);
    run<CHAINS, 5>("v_mul_e32", w, cus, d);
    run<CHAINS, 6>("v_add_e32", w, cus, d);
    run<CHAINS, 7>("v_fma 3vgpr", w, cus, d);
    run<CHAINS, 8>("v_xor_e32", w, cus, d);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    float* d;
    hipMalloc(&d, 64);
    const int cus = p.multiProcessorCount;
    for (int i = 0; i < 40; ++i) k_pk<8, 3><<<cus * 8, 256>>>(d, 4096, 1.0001f, 0.5f);  // clock warm-up
    hipDeviceSynchronize();
    for (int w : {1, 6})
    {
        row<2>(w, cus, d);
        row<8>(w, cus, d);
    }
    return 0;
}
