// Microbenchmark: the CEILING of an f32-MFMA formulation of the headline kernel's pair tests, measured
// before any of it is built.  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/mfma_coissue.hip -o /tmp/mc && /tmp/mc
//
// Today a 64-ray chunk of k_trace_shade executes 1638 vector instructions: 16 fan pairs x 66 (= 1056) for the exhaustive
// closest hit and ~580 of shading.  In Pluecker form (det, U, V, T) of a triangle are linear in x = (d, o x d, o, 1): a chunk is
// C[64 rays x 128 outputs] = X[64 x 12] W[12 x 128], i.e. 48 v_mfma_f32_32x32x2_f32 or 96 v_mfma_f32_16x16x4_f32 (12 of the 48 products
// per output multiply structural zeros; the f32 MFMA runs at the vector rate: MI355X_MICROARCH.md), after which the inside tests,
// the candidate t and the (t, id) minimum still run on the vector ALU: ~20 instructions per 16-ray x 4-triangle tile (32 tiles),
// the operand transposes through LDS and the cross-quarter reductions: ~740 instead of 1056.
//
// The kernels below have NO data flow of the real thing; they have its instruction COUNTS per chunk, on six waves per SIMD like the
// product kernel, and answer one question: if everything else were free, what would the chunk cost?
//   valu1638      1638 dependent-chain v_fma_f32 per chunk                         (today)
//   valu1322      1322                                                             (what is left beside the MFMAs)
//   mfma32 / 16   1322 v_fma_f32 + 48 v_mfma_f32_32x32x2_f32 / 96 v_mfma_f32_16x16x4_f32 per chunk, interleaved evenly
//   mfma_only     the MFMAs alone
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// MODE 0: VALU only (NV instructions per chunk); 1: + 48 x 32x32x2; 2: + 96 x 16x16x4; 3 / 4: those MFMAs alone
template <int MODE, int NV>
__global__ __launch_bounds__(256, 6) void k_mix(float* out, int chunks, float a, float b)
{
    const long long c0 = clock64(), w0 = wall_clock64();
    float x[8];
    for (int c = 0; c < 8; ++c) x[c] = threadIdx.x * 0.001f + c;
    v16f acc32[2];
    v4f  acc16[4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
    for (int i = 0; i < 4; ++i) acc16[i] = v4f{0.f, 0.f, 0.f, 0.f};
    const float ma = threadIdx.x * 0.5f, mb = a;
    // VGPR sources, like the product kernel's arithmetic (a v_fma_f32 with SGPR sources issues slower: tools/micro/pk_peak.hip)
    const float ya = a + threadIdx.x * 1e-9f, yb = b + threadIdx.x * 1e-9f;
    constexpr int NM    = (MODE == 1 || MODE == 3) ? 48 : ((MODE == 2 || MODE == 4) ? 96 : 0);
    constexpr int VPM   = (MODE == 1 || MODE == 2) ? NV / NM : 0;             // vector instructions between two MFMAs
    constexpr int VREST = (MODE == 1 || MODE == 2) ? NV - VPM * NM : (MODE == 0 ? NV : 0);
    for (int ch = 0; ch < chunks; ++ch)
    {
#pragma unroll
        for (int m = 0; m < NM; ++m)
        {
            if (MODE == 1 || MODE == 3) acc32[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc32[m & 1], 0, 0, 0);
            if (MODE == 2 || MODE == 4) acc16[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc16[m & 3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < VPM; ++v) x[v & 7] = __builtin_fmaf(x[v & 7], ya, yb);
        }
#pragma unroll 16
        for (int v = 0; v < VREST; ++v) x[v & 7] = __builtin_fmaf(x[v & 7], ya, yb);
    }
    float s = 0.f;
    for (int c = 0; c < 8; ++c) s += x[c];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 16; ++j) s += acc32[i][j];
    for (int i = 0; i < 4; ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        ((long long*)out)[1] = clock64() - c0;
        ((long long*)out)[2] = wall_clock64() - w0;
    }
}

template <int MODE, int NV>
double run(const char* name, int cus, float* d)
{
    const int  chunks = 512, blocks_per_cu = 6;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k_mix<MODE, NV><<<cus * blocks_per_cu, 256>>>(d, 8, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_mix<MODE, NV><<<cus * blocks_per_cu, 256>>>(d, chunks, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long t[3];
    hipMemcpy(t, d, sizeof(t), hipMemcpyDeviceToHost);
    const double ghz = (double)t[1] / ((double)t[2] * 10.0);
    // six waves per SIMD run `chunks` chunks each: SIMD cycles per chunk = time x clock / (6 x chunks)
    const double cyc = ms * 1e-3 * ghz * 1e9 / (6.0 * chunks);
    printf("%-34s %.3f ms, clock %.2f GHz, %.0f SIMD cycles per chunk\n", name, ms, ghz, cyc);
    return cyc;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    float* d;
    hipMalloc(&d, 64);
    const int cus = p.multiProcessorCount;
    for (int i = 0; i < 20; ++i) k_mix<0, 1638><<<cus * 6, 256>>>(d, 256, 1.0001f, 0.5f);  // clock warm-up
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep)
    {
        const double t0 = run<0, 1638>("valu1638 (today)", cus, d);
        const double t1 = run<0, 1322>("valu1322", cus, d);
        const double t2 = run<1, 1322>("valu1322 + 48 x mfma 32x32x2 f32", cus, d);
        const double t3 = run<2, 1322>("valu1322 + 96 x mfma 16x16x4 f32", cus, d);
        run<3, 0>("48 x mfma 32x32x2 f32 alone", cus, d);
        run<4, 0>("96 x mfma 16x16x4 f32 alone", cus, d);
        printf("  ceiling of the matrix-pipe form: chunk x %.3f (32x32x2), x %.3f (16x16x4) of today's; without any MFMA x %.3f\n", t2 / t0, t3 / t0,
               t1 / t0);
    }
    return 0;
}
