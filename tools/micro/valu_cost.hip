// valu_cost.hip -- shader cycles per wave64 instruction per SIMD on MI355X for the instruction CLASSES the traversal and reconstruction
// kernels are made of: the currency the kernels' instruction counts are converted with.
//
// Round 6 rewrite (docs/experiments.md (74)).  The round-5 version timed with hipEvents and converted with an ASSUMED
// 2.4 GHz and an ASSUMED one-workgroup-per-CU placement, and its one-instruction asm statements carried "vcc", "s10", "s11" clobbers -- for
// which hipcc puts an `s_nop 0` between every two statements (4 issue cycles each for a wave alone).  Its "8.1 cycles for one wave, 3.0 for
// six" were the instruction PLUS a nop.  Now:
//   * every timed block is ONE asm statement of 128 instructions (8 independent chains x 16), so nothing can be inserted between them
//     (tests/test_micro_build.py checks the listing: no s_nop inside a timed loop);
//   * cycles come from s_memtime stamps around the loop (per wave; the median over waves is printed), the clock the kernel really ran at
//     from s_memtime / s_memrealtime (100 MHz), nothing is assumed;
//   * every wave records HW_ID and XCC_ID, the census printed beside every row says how many waves really shared a SIMD.
// Columns: cycles per instruction per SIMD (span from the SIMD's first wave's start to its last wave's end / instructions of its waves: the
// throughput figure) and cycles per instruction of one wave's own stream (its share of the SIMD: uneven, the oldest wave wins).
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_cost.hip -o gpurun_out/valu_cost && gpurun_out/valu_cost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CH8(I) I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")
#define R16(S) S S S S S S S S S S S S S S S S

// instruction texts; x = the chain's register (read and written), %8 / %9 = two loop-invariant vector operands
#define I_FMA(x) "v_fma_f32 " x ", " x ", %8, %9\n\t"
#define I_FMAC(x) "v_fmac_f32_e32 " x ", %8, %9\n\t"
#define I_MUL_LIT(x) "v_mul_f32_e32 " x ", 0x3e991687, " x "\n\t"
#define I_FMA_ABS(x) "v_fma_f32 " x ", |" x "|, %8, %9\n\t"
#define I_ADD_F32(x) "v_add_f32_e32 " x ", " x ", %8\n\t"
#define I_CVT_UB0(x) "v_cvt_f32_ubyte0_e32 " x ", " x "\n\t"
#define I_CVT_UB3(x) "v_cvt_f32_ubyte3_e32 " x ", " x "\n\t"
#define I_CVT_U32(x) "v_cvt_f32_u32_e32 " x ", " x "\n\t"
#define I_MAX3(x) "v_max3_f32 " x ", " x ", %8, %9\n\t"
#define I_MIN3(x) "v_min3_f32 " x ", " x ", %8, %9\n\t"
#define I_MED3(x) "v_med3_f32 " x ", " x ", %8, %9\n\t"
#define I_MIN(x) "v_min_f32_e32 " x ", " x ", %8\n\t"
#define I_CMP_CND_VCC(x) "v_cmp_le_f32_e32 vcc, " x ", %8\n\tv_cndmask_b32_e32 " x ", " x ", %9, vcc\n\t"
#define I_CND_SGPR(x) "v_cndmask_b32_e64 " x ", " x ", %8, s[10:11]\n\t"
#define I_CMP_SGPR(x) "v_cmp_le_f32_e64 s[10:11], " x ", %8\n\t"
#define I_PERM(x) "v_perm_b32 " x ", " x ", %8, %9\n\t"
#define I_BFE(x) "v_bfe_u32 " x ", " x ", 3, 8\n\t"
#define I_AND_OR(x) "v_and_or_b32 " x ", " x ", %8, %9\n\t"
#define I_LSHL_OR(x) "v_lshl_or_b32 " x ", " x ", 1, %8\n\t"
#define I_ADD_U32(x) "v_add_u32_e32 " x ", " x ", %8\n\t"
#define I_MUL_LO(x) "v_mul_lo_u32 " x ", " x ", %8\n\t"
#define I_MOV(x) "v_mov_b32_e32 " x ", %8\n\t"
#define I_FFBH(x) "v_ffbh_u32_e32 " x ", " x "\n\t"
#define I_BCNT(x) "v_bcnt_u32_b32 " x ", %8, " x "\n\t"
#define I_BITOP3(x) "v_bitop3_b32 " x ", " x ", %8, %9 bitop3:0x6c\n\t"
#define I_EXP(x) "v_exp_f32_e32 " x ", " x "\n\t"
#define I_LOG(x) "v_log_f32_e32 " x ", " x "\n\t"
#define I_RCP(x) "v_rcp_f32_e32 " x ", " x "\n\t"
#define I_RSQ(x) "v_rsq_f32_e32 " x ", " x "\n\t"
#define I_SQRT(x) "v_sqrt_f32_e32 " x ", " x "\n\t"
#define I_PK_FMA(x) "v_pk_fma_f32 " x ", " x ", %8, %9\n\t"
#define I_PK_MUL(x) "v_pk_mul_f32 " x ", " x ", %8\n\t"
#define I_ADDC(x) "v_addc_co_u32_e32 " x ", vcc, " x ", " x ", vcc\n\t"
#define I_DS_NOP(x) "s_nop 0\n\t"

// id, name, instruction macro, operand kind (0: float chains, 1: packed-float chains), instructions per chain step, clobbers
#define OP_LIST(X)                                              \
    X(0, "v_fma_f32", I_FMA, 0, 1)                              \
    X(1, "v_fmac_f32_e32", I_FMAC, 0, 1)                        \
    X(2, "v_mul_f32 (literal)", I_MUL_LIT, 0, 1)                \
    X(3, "v_fma_f32 (|abs| source)", I_FMA_ABS, 0, 1)           \
    X(4, "v_add_f32_e32", I_ADD_F32, 0, 1)                      \
    X(5, "v_cvt_f32_ubyte0", I_CVT_UB0, 0, 1)                   \
    X(6, "v_cvt_f32_ubyte3", I_CVT_UB3, 0, 1)                   \
    X(7, "v_cvt_f32_u32", I_CVT_U32, 0, 1)                      \
    X(8, "v_max3_f32", I_MAX3, 0, 1)                            \
    X(9, "v_min3_f32", I_MIN3, 0, 1)                            \
    X(10, "v_med3_f32", I_MED3, 0, 1)                           \
    X(11, "v_min_f32_e32", I_MIN, 0, 1)                         \
    X(12, "v_cmp_le_f32 vcc + v_cndmask vcc (pair)", I_CMP_CND_VCC, 0, 2) \
    X(13, "v_cndmask_b32_e64 (sgpr mask)", I_CND_SGPR, 0, 1)    \
    X(14, "v_cmp_le_f32_e64 (sgpr dst)", I_CMP_SGPR, 0, 1)      \
    X(15, "v_perm_b32", I_PERM, 0, 1)                           \
    X(16, "v_bfe_u32", I_BFE, 0, 1)                             \
    X(17, "v_and_or_b32", I_AND_OR, 0, 1)                       \
    X(18, "v_lshl_or_b32", I_LSHL_OR, 0, 1)                     \
    X(19, "v_add_u32_e32", I_ADD_U32, 0, 1)                     \
    X(20, "v_mul_lo_u32", I_MUL_LO, 0, 1)                       \
    X(21, "v_mov_b32", I_MOV, 0, 1)                             \
    X(22, "v_ffbh_u32", I_FFBH, 0, 1)                           \
    X(23, "v_bcnt_u32_b32", I_BCNT, 0, 1)                       \
    X(24, "v_bitop3_b32", I_BITOP3, 0, 1)                       \
    X(25, "v_addc_co_u32 (vcc in/out)", I_ADDC, 0, 1)           \
    X(26, "v_exp_f32", I_EXP, 0, 1)                             \
    X(27, "v_log_f32", I_LOG, 0, 1)                             \
    X(28, "v_rcp_f32", I_RCP, 0, 1)                             \
    X(29, "v_rsq_f32", I_RSQ, 0, 1)                             \
    X(30, "v_sqrt_f32", I_SQRT, 0, 1)                           \
    X(31, "v_pk_fma_f32", I_PK_FMA, 1, 1)                       \
    X(32, "v_pk_mul_f32", I_PK_MUL, 1, 1)                       \
    X(33, "s_nop 0", I_DS_NOP, 0, 1)

struct Stamp
{
    uint64_t t0, t1, real;  // s_memtime before / after the timed loop, s_memrealtime delta around it
    uint32_t hw_id, xcc_id;
};

// EXEC_MODE: 0 all lanes, 1 lanes 0..31, 2 even lanes, 3 lanes 0..15 (docs/experiments.md (62): a half-empty EXEC costs the same)
template <int OP, int EXEC_MODE>
__global__ __launch_bounds__(256) void k_op(Stamp* out, float* sink, int iters, float a, float b)
{
    float x0 = 1.0f + threadIdx.x * 0.001f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    v2f   p0 = {x0, x0 + .5f}, p1 = {x1, x1 + .5f}, p2 = {x2, x2 + .5f}, p3 = {x3, x3 + .5f}, p4 = {x4, x4 + .5f}, p5 = {x5, x5 + .5f},
        p6 = {x6, x6 + .5f}, p7 = {x7, x7 + .5f};
    const v2f      a2 = {a, a}, b2 = {b, b};
    const unsigned lane = threadIdx.x & 63u;
    const bool     on = EXEC_MODE == 0 ? true : EXEC_MODE == 1 ? lane < 32u : EXEC_MODE == 2 ? (lane & 1u) == 0u : lane < 16u;
    uint64_t       t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    if (on)
    {
        r0 = __builtin_amdgcn_s_memrealtime();
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i)
        {
#define X(id, name, I, kind, n)                                                                                                          \
    if constexpr (OP == id)                                                                                                              \
    {                                                                                                                                    \
        if constexpr (kind == 0)                                                                                                         \
            asm volatile(R16(CH8(I)) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc", "s10", "s11"); \
        else                                                                                                                             \
            asm volatile(R16(CH8(I)) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(a2), "v"(b2)); \
    }
            OP_LIST(X)
#undef X
        }
        t1 = __builtin_amdgcn_s_memtime();
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x;
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (lane == 0)
    {
        Stamp st;
        st.t0 = t0, st.t1 = t1, st.real = r1 - r0;
        st.hw_id  = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID: wave [3:0] simd [5:4] pipe [7:6] cu [11:8] sh [12] se [15:13]
        st.xcc_id = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID [3:0]
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = st;
    }
}

template <int OP, int EXEC_MODE = 0>
void run(const char* name, int n_inst, int per_cu, int cus, Stamp* d, float* sink, const char* suffix = "")
{
    const int iters = 1024, waves = cus * per_cu * 4;
    k_op<OP, EXEC_MODE><<<cus * per_cu, 256>>>(d, sink, 8, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    k_op<OP, EXEC_MODE><<<cus * per_cu, 256>>>(d, sink, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    std::vector<Stamp> h(waves);
    hipMemcpy(h.data(), d, sizeof(Stamp) * waves, hipMemcpyDeviceToHost);
    // census: waves per (xcc, se, sh, cu, simd); per SIMD, the span from its first wave's start to its last wave's end -- the waves of a
    // SIMD are not served evenly (the oldest wins the arbitration), so a wave's own stream time says little about the SIMD's rate
    struct Simd
    {
        int      waves = 0;
        uint64_t first = ~0ull, last = 0;
    };
    std::map<uint32_t, Simd> simd;
    for (auto& s : h)
    {
        Simd& q = simd[((s.xcc_id & 0xfu) << 16) | (s.hw_id & 0xff30u)];
        q.waves++, q.first = std::min(q.first, s.t0), q.last = std::max(q.last, s.t1);
    }
    std::vector<int>    occ;
    std::vector<double> per_simd;
    for (auto& kv : simd)
    {
        occ.push_back(kv.second.waves);
        per_simd.push_back((double)(kv.second.last - kv.second.first) / ((double)kv.second.waves * iters * 128 * n_inst));
    }
    std::sort(occ.begin(), occ.end()), std::sort(per_simd.begin(), per_simd.end());
    std::vector<double> cyc, clk;
    for (auto& s : h) cyc.push_back((double)(s.t1 - s.t0) / ((double)iters * 128 * n_inst)), clk.push_back((double)(s.t1 - s.t0) / (double)s.real * 0.1);
    std::sort(cyc.begin(), cyc.end()), std::sort(clk.begin(), clk.end());
    printf("%-40s%-12s %d wg/CU asked | waves per SIMD min %d median %d max %d on %zu SIMDs | per SIMD %5.2f cyc/inst (p10 %.2f p90 %.2f) | one wave's stream %6.2f (p10 %.2f p90 %.2f) | clock %.2f GHz\n",
           name, suffix, per_cu, occ.front(), occ[occ.size() / 2], occ.back(), occ.size(), per_simd[per_simd.size() / 2], per_simd[per_simd.size() / 10],
           per_simd[per_simd.size() * 9 / 10], cyc[cyc.size() / 2], cyc[cyc.size() / 10], cyc[cyc.size() * 9 / 10], clk[clk.size() / 2]);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    Stamp* d;
    float* sink;
    hipMalloc(&d, sizeof(Stamp) * p.multiProcessorCount * 8 * 4);
    hipMalloc(&sink, 4096);
    for (int i = 0; i < 40; ++i) k_op<0, 0><<<p.multiProcessorCount * 8, 256>>>(d, sink, 1024, 1.0001f, 0.5f);  // settle the clock under load
    hipDeviceSynchronize();
    printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
    for (int per_cu : {1, 2, 4, 6, 8})
    {
#define X(id, name, I, kind, n) run<id>(name, n, per_cu, p.multiProcessorCount, d, sink);
        OP_LIST(X)
#undef X
    }
    // docs/experiments.md (62): does an instruction cost less with part of EXEC empty?
    for (int per_cu : {1, 4})
    {
        run<0, 0>("v_fma_f32", 1, per_cu, p.multiProcessorCount, d, sink, " all 64");
        run<0, 1>("v_fma_f32", 1, per_cu, p.multiProcessorCount, d, sink, " lanes 0..31");
        run<0, 2>("v_fma_f32", 1, per_cu, p.multiProcessorCount, d, sink, " even lanes");
        run<0, 3>("v_fma_f32", 1, per_cu, p.multiProcessorCount, d, sink, " lanes 0..15");
    }
    return 0;
}
