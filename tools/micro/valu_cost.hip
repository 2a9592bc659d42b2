// valu_cost.hip -- cycles per wave64 instruction per SIMD on MI355X for the instruction CLASSES the traversal and reconstruction
// kernels are made of, at 1 and 6 waves per SIMD: the currency the kernels' instruction counts are converted with.
// (tools/micro/pk_peak.hip measured v_fma_f32 / packed forms; this adds conversions, three-operand min / max, selects, compares,
// permutes, bit-field ops, literals, source modifiers and the transcendentals.)
// Each op runs in 8 independent chains per lane, 16 x unrolled, inline assembly so that the opcode is the one named.
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_cost.hip -o gpurun_out/valu_cost && gpurun_out/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP_LIST(X)                                                                                              \
    X(0, "v_fma_f32 (3 vgpr)", "v_fma_f32 %0, %0, %1, %2")                                                      \
    X(1, "v_fmac_f32_e32", "v_fmac_f32_e32 %0, %1, %2")                                                         \
    X(2, "v_mul_f32 literal", "v_mul_f32_e32 %0, 0x3e991687, %0")                                               \
    X(3, "v_fma_f32 |abs| src", "v_fma_f32 %0, |%0|, %1, %2")                                                   \
    X(4, "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0_e32 %0, %1")                                                     \
    X(5, "v_cvt_f32_ubyte3", "v_cvt_f32_ubyte3_e32 %0, %1")                                                     \
    X(6, "v_max3_f32", "v_max3_f32 %0, %0, %1, %2")                                                             \
    X(7, "v_min_f32_e32", "v_min_f32_e32 %0, %0, %1")                                                           \
    X(8, "v_cmp_le_f32 + v_cndmask (vcc)", "v_cmp_le_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc") \
    X(9, "v_cndmask_b32_e64 (sgpr mask)", "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")                             \
    X(10, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2")                                                            \
    X(11, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 8")                                                                \
    X(12, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2")                                                        \
    X(13, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 1, %1")                                                       \
    X(14, "v_add_u32_e32", "v_add_u32_e32 %0, %0, %1")                                                          \
    X(15, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1")                                                            \
    X(16, "v_exp_f32", "v_exp_f32_e32 %0, %0")                                                                  \
    X(17, "v_log_f32", "v_log_f32_e32 %0, %0")                                                                  \
    X(18, "v_rcp_f32", "v_rcp_f32_e32 %0, %0")                                                                  \
    X(19, "v_rsq_f32", "v_rsq_f32_e32 %0, %0")                                                                  \
    X(20, "v_sqrt_f32", "v_sqrt_f32_e32 %0, %0")                                                                \
    X(21, "v_pk_fma_f32", "v_pk_fma_f32 %3, %3, %4, %5")                                                        \
    X(22, "v_pk_mul_f32", "v_pk_mul_f32 %3, %3, %4")                                                            \
    X(23, "v_cvt_f32_u32", "v_cvt_f32_u32_e32 %0, %1")                                                          \
    X(24, "v_min3_f32", "v_min3_f32 %0, %0, %1, %2")                                                            \
    X(25, "v_med3_f32", "v_med3_f32 %0, %0, %1, %2")                                                            \
    X(26, "v_cmp_le_f32_e64 (sgpr dst)", "v_cmp_le_f32_e64 s[10:11], %0, %1")                                   \
    X(27, "v_mov_b32", "v_mov_b32_e32 %0, %1")                                                                  \
    X(28, "v_ffbh_u32", "v_ffbh_u32_e32 %0, %1")                                                                \
    X(29, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %1, %0")                                                        \
    X(30, "v_div_scale+fixup pair", "v_div_scale_f32 %0, vcc, %0, %1, %0\n\tv_div_fixup_f32 %0, %0, %1, %2")

typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k_op(float* out, int iters, float a, float b)
{
    float x[8];
    v2f   p[8];
    for (int c = 0; c < 8; ++c) x[c] = 1.0f + threadIdx.x * 0.001f + c, p[c] = v2f{x[c], x[c] + 0.5f};
    const v2f a2 = {a, a}, b2 = {b, b};
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < 8; ++c)
            {
#define X(id, name, text) \
    if (OP == id) asm volatile(text : "+v"(x[c]) : "v"(a), "v"(b), "v"(p[c]), "v"(a2), "v"(b2) : "vcc", "s10", "s11");
                OP_LIST(X)
#undef X
            }
    }
    float s = 0.f;
    for (int c = 0; c < 8; ++c) s += x[c] + p[c].x;
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
void run(const char* name, int per_cu, int cus, float* d, int n_inst)
{
    const int  iters = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k_op<OP><<<cus * per_cu, 256>>>(d, 16, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_op<OP><<<cus * per_cu, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)per_cu * iters * 16 * 8 * n_inst;  // wave instructions per SIMD
    printf("%-34s %d waves/SIMD  %8.3f ms  %6.2f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", name, per_cu, ms,
           ms * 1e-3 * 2.4e9 / inst);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    float* d;
    hipMalloc(&d, 4096);
    for (int i = 0; i < 20; ++i) k_op<0><<<p.multiProcessorCount * 8, 256>>>(d, 2048, 1.0001f, 0.5f);  // clock warm-up
    hipDeviceSynchronize();
    for (int per_cu : {1, 6})
    {
#define X(id, name, text) run<id>(name, per_cu, p.multiProcessorCount, d, (id == 8 || id == 30) ? 2 : 1);
        OP_LIST(X)
#undef X
    }
    return 0;
}
