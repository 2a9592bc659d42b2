// wave_chase.hip -- what does a WAVE-UNIFORM dependent record fetch cost on an MI355X, by the path it takes?
//
// The camera rays' packet walk (csrc/kernels.hip traverse_closest_packet) fetches one 64-byte node per wave and step through the
// scalar cache and then depends on what arrived; its vector ALU is ~0.2 busy at eight waves per SIMD and its time falls far less than
// linearly with the waves per SIMD (docs/experiments.md (84)).  Latency, or the throughput of the scalar cache's miss path?  Every
// wave here runs one pointer chase (next record = hash of a word of the record just fetched) through a table of 64-byte records:
//   scalar    one s_load_dwordx16 per step (what the walk does)
//   scalar2   two independent chases per wave, both s_loads in flight together (what speculation / two packets would do)
//   vec4      lanes 0..3 fetch the record's four 16-byte pieces with ONE global_load_dwordx4 (EXEC = 0xf), v_readlane to SGPRs
//   vec8      the same for two records per step (lanes 0..3 and 4..7, one instruction), two chases per wave
//   vecall    all 64 lanes fetch the same 16 bytes x 4 instructions (the per-lane form the walk replaced in round 2)
// with W waves per SIMD resident on every CU (dynamic LDS caps the residency) and `work` dependent FMAs per step between the fetch
// and the next address (the walk's slab tests).  One JSON line per configuration.
//
// hipcc --offload-arch=gfx950 -O3 tools/micro/wave_chase.hip -o gpurun_out/wave_chase
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                   \
    do                                                                                             \
    {                                                                                              \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess)                                                                      \
        {                                                                                          \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
            exit(2);                                                                               \
        }                                                                                          \
    } while (0)

enum Mode { SCALAR, SCALAR2, VEC4, VEC8, VECALL, N_MODES };
static const char* kModeName[N_MODES] = {"scalar", "scalar2", "vec4", "vec8", "vecall"};

struct Rec
{
    uint32_t w[16];
};

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t reduce(uint32_t h, uint32_t n) { return (uint32_t)(((uint64_t)h * n) >> 32); }

__global__ void k_fill(uint32_t* t, size_t words)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
        t[i] = mix((uint32_t)i * 2654435761u + 12345u);
}

// one 64-byte record through the scalar cache (constant address space), all sixteen words used
__device__ __forceinline__ void load_scalar(const Rec* table, uint32_t i, uint32_t& x, uint32_t& a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const Rec ConstRec;
    const Rec r = ((const ConstRec*)table)[i];
#else
    const Rec r = table[i];
#endif
    x = 0u;
    for (int k = 0; k < 16; ++k) x ^= r.w[k];
    a = r.w[3];
}

// `work` dependent FMAs on a value derived from the record, folded back into the next index (so they sit on the chain)
__device__ __forceinline__ uint32_t chain_work(uint32_t x, uint32_t work)
{
    float f = __uint_as_float((x & 0x007fffffu) | 0x3f800000u);
    for (uint32_t k = 0; k < work; ++k) f = __builtin_fmaf(f, 0.999f, 0.001f);
    return x ^ (__float_as_uint(f) & 1u);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_chase(const Rec* __restrict__ table, uint32_t n_records, uint32_t steps, uint32_t work, uint32_t* out,
                                                uint32_t seed)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane   = threadIdx.x & 63u;
    const uint32_t wave_g = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t       i0 = __builtin_amdgcn_readfirstlane(reduce(mix(wave_g * 2u + seed), n_records));
    uint32_t       i1 = __builtin_amdgcn_readfirstlane(reduce(mix(wave_g * 2u + 1u + seed), n_records));
    uint32_t       acc = 0u;
    for (uint32_t s = 0; s < steps; ++s)
    {
        if (MODE == SCALAR)
        {
            uint32_t x, a;
            load_scalar(table, i0, x, a);
            x = chain_work(x, work);
            acc += a;
            i0 = __builtin_amdgcn_readfirstlane(reduce(mix(x), n_records));
        }
        else if (MODE == SCALAR2)
        {
            uint32_t x, y, a, b;
            load_scalar(table, i0, x, a);
            load_scalar(table, i1, y, b);
            x = chain_work(x, work), y = chain_work(y, work);
            acc += a + b;
            i0 = __builtin_amdgcn_readfirstlane(reduce(mix(x), n_records));
            i1 = __builtin_amdgcn_readfirstlane(reduce(mix(y), n_records));
        }
        else if (MODE == VEC4 || MODE == VEC8)
        {
            // lane l < 4 (8): piece l & 3 of record i0 (i1 for lanes 4..7)
            const uint32_t idx = (MODE == VEC8 && lane >= 4u) ? i1 : i0;
            uint4          v   = make_uint4(0u, 0u, 0u, 0u);
            if (lane < (MODE == VEC8 ? 8u : 4u)) v = ((const uint4*)(table + idx))[lane & 3u];
            uint32_t x = __builtin_amdgcn_readlane(v.x, 0) ^ __builtin_amdgcn_readlane(v.y, 1) ^ __builtin_amdgcn_readlane(v.z, 2) ^
                         __builtin_amdgcn_readlane(v.w, 3);
            x = chain_work(x, work);
            acc += __builtin_amdgcn_readlane(v.w, 0);
            i0 = __builtin_amdgcn_readfirstlane(reduce(mix(x), n_records));
            if (MODE == VEC8)
            {
                uint32_t y = __builtin_amdgcn_readlane(v.x, 4) ^ __builtin_amdgcn_readlane(v.y, 5) ^ __builtin_amdgcn_readlane(v.z, 6) ^
                             __builtin_amdgcn_readlane(v.w, 7);
                y = chain_work(y, work);
                acc += __builtin_amdgcn_readlane(v.w, 4);
                i1 = __builtin_amdgcn_readfirstlane(reduce(mix(y), n_records));
            }
        }
        else  // VECALL
        {
            const uint4* p = (const uint4*)(table + i0);
            const uint4  a = p[0], b = p[1], c = p[2], d = p[3];
            uint32_t     x = a.x ^ b.y ^ c.z ^ d.w;
            x              = chain_work(x, work);
            acc += a.w;
            i0 = __builtin_amdgcn_readfirstlane(reduce(mix(x), n_records));
        }
    }
    if (lane == 0) out[wave_g] = acc + i0 + i1 + lds[0] * 0u;
}

template <int MODE>
static void run(const Rec* table, uint32_t n_records, uint32_t steps, uint32_t work, int waves_per_simd, int cus, uint32_t* out, const char* tname)
{
    // W waves per SIMD = W workgroups of four waves per CU; the LDS request leaves room for exactly W of them (160 KB per CU)
    const size_t lds_bytes = (size_t)(160 * 1024 / waves_per_simd) - 1024;
    CHECK(hipFuncSetAttribute((const void*)k_chase<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const uint32_t grid = (uint32_t)(cus * waves_per_simd);
    hipEvent_t     e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chase<MODE>, dim3(grid), dim3(256), lds_bytes, 0, table, n_records, steps / 8, work, out, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chase<MODE>, dim3(grid), dim3(256), lds_bytes, 0, table, n_records, steps, work, out, 7u + rep);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const int    chains   = (MODE == SCALAR2 || MODE == VEC8) ? 2 : 1;
    const double ns_step  = best * 1e6 / steps;                                   // one wave's step (both chains of a two-chain mode)
    const double per_cu   = (double)waves_per_simd * 4 * chains / ns_step * 1e3;  // records per microsecond and CU
    printf("{\"mode\": \"%s\", \"table\": \"%s\", \"waves_per_simd\": %d, \"work_fmas\": %u, \"ns_per_step\": %.1f, \"records_per_us_per_cu\": %.1f}\n",
           kModeName[MODE], tname, waves_per_simd, work, ns_step, per_cu);
    fflush(stdout);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv)
{
    uint32_t steps = 20000;
    for (int i = 1; i < argc; ++i)
        if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = (uint32_t)atoi(argv[++i]);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t max_bytes = (size_t)32 << 20;
    Rec*         table     = nullptr;
    uint32_t*    out       = nullptr;
    CHECK(hipMalloc(&table, max_bytes));
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 4 * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (uint32_t*)table, max_bytes / 4);
    CHECK(hipDeviceSynchronize());
    struct T { const char* name; size_t bytes; } tables[] = {{"1MB", (size_t)1 << 20}, {"32MB", max_bytes}};
    const int      ws[]    = {1, 2, 4, 8};
    const uint32_t works[] = {0u, 16u};
    for (const T& t : tables)
        for (uint32_t work : works)
            for (int w : ws)
            {
                const uint32_t n = (uint32_t)(t.bytes / sizeof(Rec));
                run<SCALAR>(table, n, steps, work, w, cus, out, t.name);
                run<SCALAR2>(table, n, steps, work, w, cus, out, t.name);
                run<VEC4>(table, n, steps, work, w, cus, out, t.name);
                run<VEC8>(table, n, steps, work, w, cus, out, t.name);
                if (work == 0u) run<VECALL>(table, n, steps, work, w, cus, out, t.name);
            }
    CHECK(hipFree(table));
    CHECK(hipFree(out));
    return 0;
}
