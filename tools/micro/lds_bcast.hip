// Microbenchmark: wave-uniform operands of a VALU loop from SGPRs (s_load) versus from VGPRs filled by LDS broadcast reads
// (ds_read_b128 at a wave-uniform address).  Models the fan-pair loop: per iteration 20 uniform floats, 24 fma that use them and
// 42 fma that do not.
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/lds_bcast.hip -o /tmp/lds_bcast && /tmp/lds_bcast
#include <hip/hip_runtime.h>
#include <cstdio>

struct alignas(16) Rec
{
    float f[20];
};

template <int MODE>  // 0: SGPR operands (constant address space), 1: LDS broadcast reads, 2: no uniform operands at all
__global__ __launch_bounds__(256, 6) void k(const Rec* recs, int n_rec, int iters, float* out)
{
    __shared__ float4 lds[5 * 64];
    for (int i = threadIdx.x; i < 5 * n_rec; i += 256) lds[i] = reinterpret_cast<const float4*>(recs)[i];
    __syncthreads();
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f;
    for (int it = 0; it < iters; ++it)
        for (int r = 0; r < n_rec; ++r)
        {
            float u[20];
            if (MODE == 0)
            {
#if defined(__HIP_DEVICE_COMPILE__)
                typedef __attribute__((address_space(4))) const Rec CRec;
                const Rec v = ((const CRec*)recs)[r];
                for (int i = 0; i < 20; ++i) u[i] = v.f[i];
#else
                for (int i = 0; i < 20; ++i) u[i] = recs[r].f[i];
#endif
            }
            else if (MODE == 1)
            {
                for (int i = 0; i < 5; ++i)
                {
                    const float4 v = lds[5 * r + i];
                    u[4 * i] = v.x, u[4 * i + 1] = v.y, u[4 * i + 2] = v.z, u[4 * i + 3] = v.w;
                }
            }
            else
                for (int i = 0; i < 20; ++i) u[i] = a5;
            // 24 fma with a uniform operand, 42 without
#pragma unroll
            for (int i = 0; i < 24; ++i)
            {
                float& x = (i % 6 == 0) ? a0 : (i % 6 == 1) ? a1 : (i % 6 == 2) ? a2 : (i % 6 == 3) ? a3 : (i % 6 == 4) ? a4 : a5;
                x        = __builtin_fmaf(x, u[i % 20], a0 + a3);
            }
#pragma unroll
            for (int i = 0; i < 42; ++i)
            {
                float& x = (i % 6 == 0) ? a0 : (i % 6 == 1) ? a1 : (i % 6 == 2) ? a2 : (i % 6 == 3) ? a3 : (i % 6 == 4) ? a4 : a5;
                x        = __builtin_fmaf(x, a1, a2);
            }
        }
    const float s = a0 + a1 + a2 + a3 + a4 + a5;
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* name, const Rec* d, float* out, int cus)
{
    const int  n_rec = 16, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<MODE><<<cus * 6, 256>>>(d, n_rec, 10, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<cus * 6, 256>>>(d, n_rec, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 6 waves x iters x n_rec record iterations
    printf("%-28s %.3f ms, %.0f nominal (2.4 GHz) cycles per record iteration per wave-slot (66 fma)\n", name, ms,
           ms * 1e-3 * 2.4e9 / (6.0 * iters * n_rec));
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    Rec h[64];
    for (int r = 0; r < 64; ++r)
        for (int i = 0; i < 20; ++i) h[r].f[i] = 1.0f + 0.001f * (r + i);
    Rec*   d;
    float* out;
    hipMalloc(&d, sizeof(h)), hipMalloc(&out, 64);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
    {
        run<0>("SGPR operands (s_load)", d, out, p.multiProcessorCount);
        run<1>("LDS broadcast -> VGPR", d, out, p.multiProcessorCount);
        run<2>("no uniform operands", d, out, p.multiProcessorCount);
    }
    return 0;
}
