// gather_ceiling.hip -- what rate of DEPENDENT random record fetches does an MI355X sustain, by the shape of the fetch?
//
// The traversal kernels (csrc/trace8.hip) fetch one 80-byte node per lane and step as five scattered 16-byte loads and then
// depend on what arrived.  VERDICT r4 (missing 4): nothing measured what the chip can do in that access pattern, and FETCH_SIZE's
// gfx950 correction (x 2, MI355X_MICROARCH.md "HBM") is calibrated for wide coalesced streams only.  This program measures both:
// every lane runs its own pointer chase (next index = hash of a word of the record just loaded) through a table far larger than
// L2 (and, for the large table, than the Infinity Cache), in several fetch shapes:
//   lane5p   5 x 16 B per lane at an 80-B packed stride   -- the product's node fetch (a node straddles 1.5 128-B lines)
//   lane5a   5 x 16 B per lane at a 128-B stride           -- docs/experiments.md (58)
//   lane4    4 x 16 B per lane at a 64-B stride            -- (47)'s node / a triangle record
//   lane1    1 x 16 B per lane of a random 128-B line      -- the plain scattered-line ceiling of the per-lane form
//   lane8    8 x 16 B per lane, the whole 128-B line       -- bytes without lines: what the address path charges per piece
//   coop8d   8 adjacent lanes fetch ONE 128-B record with one 16-B piece each (eight LDS-DMA instructions serve the wave's 64
//            records, each writing 1 KiB = 8 whole records); the owner lane then reads its five pieces from LDS
//   coop8r   the same through registers (global_load_dwordx4 + ds_write_b128 at a conflict-free stride)
//   coop41   four adjacent lanes fetch pieces 0..3 of one record (four LDS-DMA instructions, 16 records each), a fifth instruction
//            fetches every lane's own piece 4: five instructions like lane5a, 128 instead of 320 line touches, 5 KiB of LDS per wave
//   stream   coalesced 16 B per lane over the table, no dependence: the calibration point the guide's x 2 was measured on
// Output: one JSON line per configuration on stdout; `--pmc` runs one launch per configuration so that a
// `rocprofv3 --pmc FETCH_SIZE` pass of the same command can be joined by dispatch order (tools/micro/gather_ceiling_report.py).
//
// hipcc --offload-arch=gfx950 -O3 tools/micro/gather_ceiling.hip -o gpurun_out/gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

enum Mode { LANE5P, LANE5A, LANE4, LANE1, LANE8, COOP8D, COOP8R, COOP41, STREAM, N_MODES };
static const char* kModeName[N_MODES] = {"lane5p", "lane5a", "lane4", "lane1", "lane8", "coop8d", "coop8r", "coop41", "stream"};
// record stride in 16-byte pieces, pieces loaded per record
static const int kStride16[N_MODES] = {5, 8, 4, 8, 8, 8, 8, 8, 1};
static const int kPieces[N_MODES]   = {5, 5, 4, 1, 8, 8, 8, 5, 1};

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t reduce(uint32_t h, uint32_t n) { return (uint32_t)(((uint64_t)h * n) >> 32); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

__global__ void k_fill(float4* t, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    {
        const uint32_t h = mix((uint32_t)i * 2654435761u + (uint32_t)(i >> 32));
        t[i] = make_float4(__uint_as_float(h), __uint_as_float(h * 3u), __uint_as_float(h * 5u), __uint_as_float(h * 7u));
    }
}

// Dynamic LDS: a per-wave landing zone for the coop modes (8 x 1040 B, see below) and the occupancy limiter for all of them.
template <int MODE>
__global__ __launch_bounds__(256) void k_chase(const float4* __restrict__ table, uint32_t n_records, uint32_t steps, uint32_t* out, uint32_t seed)
{
    extern __shared__ float4 lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    uint32_t       idx = reduce(mix(gid ^ seed), n_records), acc = 0u;
    constexpr int  S16 = MODE == LANE5P ? 5 : MODE == LANE4 ? 4 : 8;
    if (MODE == LANE5P || MODE == LANE5A || MODE == LANE4 || MODE == LANE1 || MODE == LANE8)
    {
        constexpr int P = MODE == LANE4 ? 4 : MODE == LANE1 ? 1 : MODE == LANE8 ? 8 : 5;
        for (uint32_t s = 0; s < steps; ++s)
        {
            const float4* p = table + (size_t)idx * S16;
            float4        v[P];
#pragma unroll
            for (int k = 0; k < P; ++k) v[k] = p[k];
            uint32_t x = 0u;
#pragma unroll
            for (int k = 0; k < P; ++k) x ^= (f2u(v[k].x) + f2u(v[k].y)) ^ (f2u(v[k].z) + f2u(v[k].w));
            acc += x;
            idx = reduce(mix(x + gid * 0x9E3779B9u + s), n_records);
        }
    }
    else if (MODE == COOP8D)
    {
        // landing zone of this wave: instruction k writes records of lanes 8k..8k+7 at byte k * 1040 (1 KiB + a 16-B skew so that the
        // owners' ds_read_b128 of a 16-lane group are at most 2-way conflicting instead of 8-way)
        float4* zone = lds + wave * (8 * 65);
        for (uint32_t s = 0; s < steps; ++s)
        {
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
                const uint32_t j   = (uint32_t)__shfl((int)idx, 8 * k + (int)(lane >> 3));
                const float4*  src = table + (size_t)j * 8 + (lane & 7u);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(zone + k * 65), 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0) (gfx9 encoding: lgkmcnt and expcnt left at their maxima)
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            const float4* own = zone + (lane >> 3) * 65 + (lane & 7u) * 8;
            uint32_t      x   = 0u;
#pragma unroll
            for (int k = 0; k < 5; ++k)
            {
                const float4 v = own[k];
                x ^= (f2u(v.x) + f2u(v.y)) ^ (f2u(v.z) + f2u(v.w));
            }
            __builtin_amdgcn_wave_barrier();
            acc += x;
            idx = reduce(mix(x + gid * 0x9E3779B9u + s), n_records);
        }
    }
    else if (MODE == COOP8R)
    {
        // record n of the wave at float4 index n * 9 (144-B stride: conflict-free 16-B reads and writes)
        float4* zone = lds + wave * (64 * 9);
        for (uint32_t s = 0; s < steps; ++s)
        {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
                const uint32_t j = (uint32_t)__shfl((int)idx, 8 * k + (int)(lane >> 3));
                v[k]             = table[(size_t)j * 8 + (lane & 7u)];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) zone[(8 * k + (lane >> 3)) * 9 + (lane & 7u)] = v[k];
            __builtin_amdgcn_wave_barrier();
            uint32_t x = 0u;
#pragma unroll
            for (int k = 0; k < 5; ++k)
            {
                const float4 o = zone[lane * 9 + k];
                x ^= (f2u(o.x) + f2u(o.y)) ^ (f2u(o.z) + f2u(o.w));
            }
            __builtin_amdgcn_wave_barrier();
            acc += x;
            idx = reduce(mix(x + gid * 0x9E3779B9u + s), n_records);
        }
    }
    else if (MODE == COOP41)
    {
        // instruction k (0..3): lane l fetches piece l & 3 of the record of lane 16 k + (l >> 2), landing record-major at a 64-B stride
        // (+ a 16-B skew per instruction); instruction 4: every lane its own piece 4
        float4* zone = lds + wave * (4 * 65 + 64);
        for (uint32_t s = 0; s < steps; ++s)
        {
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const uint32_t j   = (uint32_t)__shfl((int)idx, 16 * k + (int)(lane >> 2));
                const float4*  src = table + (size_t)j * 8 + (lane & 3u);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(zone + k * 65), 16, 0, 0);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(table + (size_t)idx * 8 + 4),
                                             (__attribute__((address_space(3))) void*)(zone + 4 * 65), 16, 0, 0);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            const float4* own = zone + (lane >> 4) * 65 + (lane & 15u) * 4;
            uint32_t      x   = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const float4 v = own[k];
                x ^= (f2u(v.x) + f2u(v.y)) ^ (f2u(v.z) + f2u(v.w));
            }
            {
                const float4 v = zone[4 * 65 + lane];
                x ^= (f2u(v.x) + f2u(v.y)) ^ (f2u(v.z) + f2u(v.w));
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            acc += x;
            idx = reduce(mix(x + gid * 0x9E3779B9u + s), n_records);
        }
    }
    else  // STREAM: `steps` coalesced 16-B loads per lane, grid-strided over the table (n_records = 16-byte pieces here)
    {
        // (the host sets steps = a multiple of 4 with steps * threads <= n_records: no wrap)
        const size_t total = (size_t)gridDim.x * 256u;
        size_t       i     = gid;
        for (uint32_t s = 0; s < steps; s += 4)
        {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = table[i + k * total];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += (f2u(v[k].x) + f2u(v[k].y)) ^ (f2u(v[k].z) + f2u(v[k].w));
            i += 4 * total;
        }
    }
    out[gid] = acc ^ idx;
}

typedef void (*KernelFn)(const float4*, uint32_t, uint32_t, uint32_t*, uint32_t);
static KernelFn kKernels[N_MODES] = {k_chase<LANE5P>, k_chase<LANE5A>, k_chase<LANE4>, k_chase<LANE1>, k_chase<LANE8>, k_chase<COOP8D>, k_chase<COOP8R>, k_chase<COOP41>, k_chase<STREAM>};

// distinct 128-B lines / 64-B sectors a record fetch touches, averaged over the table's alignment phases
static void lines_per_record(int mode, double& l128, double& s64)
{
    if (mode == STREAM) { l128 = 16.0 / 128.0, s64 = 16.0 / 64.0; return; }
    const int stride = kStride16[mode] * 16, bytes = kPieces[mode] * 16;
    double    a = 0, b = 0;
    for (int i = 0; i < 8; ++i)
    {
        const int lo = i * stride, hi = lo + bytes - 1;
        a += hi / 128 - lo / 128 + 1, b += hi / 64 - lo / 64 + 1;
    }
    l128 = a / 8, s64 = b / 8;
}

int main(int argc, char** argv)
{
    bool pmc = false, quick = false;
    for (int i = 1; i < argc; ++i) pmc |= !strcmp(argv[i], "--pmc"), quick |= !strcmp(argv[i], "--quick");
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    // 1 MB: resident in every XCD's L2 (what the address path itself sustains); 24 MB: beyond the L2s, inside the Infinity Cache (the
    // 262 k-triangle hall's 20 MB of nodes and records); 0.25 GB (the 16.8 M hall's wide nodes); 1.3 GB (+ its records)
    const int    n_tables = 4;
    const size_t table_bytes[n_tables] = {(size_t)1 << 20, (size_t)24 << 20, (size_t)256 << 20, (size_t)1331 << 20};
    float4* table;
    CHECK(hipMalloc(&table, table_bytes[n_tables - 1]));
    k_fill<<<cus * 8, 256>>>(table, table_bytes[n_tables - 1] / 16);
    uint32_t* out;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int m = 0; m < N_MODES; ++m) CHECK(hipFuncSetAttribute((const void*)kKernels[m], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    int dispatch = 0;  // k_chase launches so far: in --pmc mode a configuration is exactly ONE launch, so line n of the output is the
                       // n-th k_chase dispatch of the rocprofv3 pass (tools/micro/gather_ceiling_report.py joins them in order)
    for (int ti = 0; ti < n_tables; ++ti)
        for (int m = 0; m < N_MODES; ++m)
            for (int w = 2; w <= 8; ++w)
            {
                const bool coop8 = m == COOP8D || m == COOP8R;
                if (m == STREAM ? w != 8 : coop8 ? w > 4 : (w < 4 || w == 7)) continue;
                if ((pmc || quick) && m != STREAM && w != (coop8 ? 4 : 6)) continue;
                // blocks of 4 waves; w blocks per CU = w waves per SIMD: dynamic LDS sized so that exactly w blocks fit
                size_t lds = (size_t)(160 * 1024 / w) & ~(size_t)1023;
                const size_t need = m == COOP8D ? 4 * 8 * 65 * 16 : m == COOP8R ? 4 * 64 * 9 * 16 : m == COOP41 ? 4 * (4 * 65 + 64) * 16 : 0;
                if (lds < need) continue;
                if (w == 8) lds = need > 16384 ? need : 16384;  // eight blocks per CU: the wave limit, not LDS
                const int      grid  = cus * w;
                const uint32_t n_rec = m == STREAM ? (uint32_t)(table_bytes[ti] / 16) : (uint32_t)(table_bytes[ti] / (kStride16[m] * 16));
                uint32_t       steps = m == STREAM ? (uint32_t)(n_rec / ((size_t)grid * 256)) & ~3u : (quick ? 128u : 384u);
                if (m == STREAM && steps == 0) continue;  // (the table is smaller than one pass of the grid)
                const int      reps  = pmc ? 1 : 3;
                hipLaunchKernelGGL(kKernels[m], dim3(grid), dim3(256), lds, 0, table, n_rec, pmc ? steps : 16u, out, 1u);  // warm-up
                const int first = dispatch++;
                CHECK(hipDeviceSynchronize());
                float best = 1e30f, sum = 0.f;
                for (int r = 0; r < reps && !pmc; ++r)
                {
                    CHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(kKernels[m], dim3(grid), dim3(256), lds, 0, table, n_rec, steps, out, 7u + r);
                    ++dispatch;
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best, sum += ms;
                }
                CHECK(hipGetLastError());
                double l128, s64;
                lines_per_record(m, l128, s64);
                const double records = (double)grid * 256.0 * steps;
                const double t       = pmc ? 0.0 : best * 1e-3;
                printf("{\"mode\": \"%s\", \"table_mb\": %zu, \"waves_per_simd\": %d, \"steps\": %u, \"records_per_launch\": %.0f, "
                       "\"lines128_per_record\": %.3f, \"sectors64_per_record\": %.3f, \"payload_bytes_per_record\": %d, "
                       "\"known_bytes_lines128\": %.0f, \"known_bytes_sectors64\": %.0f, \"first_dispatch\": %d, \"pmc_mode\": %s",
                       kModeName[m], table_bytes[ti] >> 20, w, steps, records, l128, s64, kPieces[m] * 16, records * l128 * 128.0,
                       records * s64 * 64.0, first, pmc ? "true" : "false");
                if (!pmc)
                    printf(", \"ms_best\": %.4f, \"ms_avg\": %.4f, \"grecords_per_s\": %.3f, \"tbs_lines128\": %.3f, \"tbs_sectors64\": %.3f, "
                           "\"tbs_payload\": %.3f",
                           best, sum / reps, records / t / 1e9, records * l128 * 128.0 / t / 1e12, records * s64 * 64.0 / t / 1e12,
                           records * kPieces[m] * 16.0 / t / 1e12);
                printf("}\n");
                fflush(stdout);
            }
    return 0;
}
