// bvh.hip — explicit on-device LBVH build for gfx950.  Replaces the driver calls
// BuildRaytracingAccelerationStructure of BLASSystem/TLASSystem (reference src/systems/blas_system.cpp:65,
// tlas_system.cpp:72).  The reference's two-level structure (one identity-transform instance per mesh,
// InstanceID = mesh.index, tlas_system.cpp:40-58) is folded into one level; (instance, primitive) ids stay
// available per triangle.
//
// Pipeline (one stream, no host round trip): triangle setup + scene bounds -> 30-bit Morton codes ->
// LSD radix sort (4 x 8 bit, stable) -> Karras 2012 hierarchy -> bottom-up refit with arrival counters.
// The traversal result does not depend on the tree shape (closest hit = min t, ties to the lower triangle id).
#include "cap_kernels.h"
#include "cap_wide.h"

namespace cap
{
namespace
{
constexpr uint32_t kSortTile = 2048;  // elements per workgroup per radix pass

__device__ __forceinline__ uint32_t float_to_ordered(float f)
{
    const uint32_t u = f2u(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t o)
{
    return u2f((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// Triangle setup: fetch through the reference's index/vertex indirection once (scene.h:13-45), emit the
// intersection record (v0, e1, e2), the pre-gathered shading record and the triangle box; reduce scene bounds.
__global__ __launch_bounds__(kBlock) void k_tri_setup(BvhBuildArgs a)
{
    __shared__ float s_lo[kBlock / 64][3], s_hi[kBlock / 64][3];
    float4* tri_box = a.tri_box;
    float   slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
    // a fixed grid strides over the triangles: the scene bounds cost six atomics per WORKGROUP (one per wave and component was 1.6 M
    // atomics on six words at 16.8 M triangles: 17 of the kernel's 18 ms)
    for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < a.tri_count; g += gridDim.x * kBlock)
    {
        float lo[3], hi[3];
        const uint4    id  = a.tri_ids[g];
        const uint4    mo  = a.mesh_offsets[id.x];
        const uint32_t io  = mo.y + 3u * id.y;
        const uint32_t i0 = mo.x + a.indices[io], i1 = mo.x + a.indices[io + 1], i2 = mo.x + a.indices[io + 2];
        const float*   P = a.positions;
        const float*   N = a.normals;
        const float*   T = a.texcoords;
        const v3 p0 = mk3(P[3 * i0], P[3 * i0 + 1], P[3 * i0 + 2]), p1 = mk3(P[3 * i1], P[3 * i1 + 1], P[3 * i1 + 2]),
                 p2 = mk3(P[3 * i2], P[3 * i2 + 1], P[3 * i2 + 2]);
        const v3 e1 = p1 - p0, e2 = p2 - p0;
        const v3 n = cross3(e1, e2);  // plane normal of the intersection contract, evaluated once per triangle
        a.tri_raw[4 * (size_t)g + 0] = make_float4(p0.x, p0.y, p0.z, e1.x);
        a.tri_raw[4 * (size_t)g + 1] = make_float4(e1.y, e1.z, e2.x, e2.y);
        a.tri_raw[4 * (size_t)g + 2] = make_float4(e2.z, n.x, n.y, n.z);
        a.tri_raw[4 * (size_t)g + 3] = make_float4(u2f(g), 0.f, 0.f, 0.f);
        float4* st = a.shade_tris + kShadeRec * (size_t)g;
        st[0] = make_float4(p0.x, p0.y, p0.z, T[2 * i0]);
        st[1] = make_float4(p1.x, p1.y, p1.z, T[2 * i0 + 1]);
        st[2] = make_float4(p2.x, p2.y, p2.z, T[2 * i1]);
        st[3] = make_float4(N[3 * i0], N[3 * i0 + 1], N[3 * i0 + 2], T[2 * i1 + 1]);
        st[4] = make_float4(N[3 * i1], N[3 * i1 + 1], N[3 * i1 + 2], T[2 * i2]);
        st[5] = make_float4(N[3 * i2], N[3 * i2 + 1], N[3 * i2 + 2], T[2 * i2 + 1]);
        st[6] = make_float4(u2f(id.x), u2f(id.y), u2f(id.z), 0.f);  // (instance, primitive, texture index): rides in the record's second sector
        st[7] = make_float4(0.f, 0.f, 0.f, 0.f);
        lo[0] = fminf(p0.x, fminf(p1.x, p2.x)), lo[1] = fminf(p0.y, fminf(p1.y, p2.y)), lo[2] = fminf(p0.z, fminf(p1.z, p2.z));
        hi[0] = fmaxf(p0.x, fmaxf(p1.x, p2.x)), hi[1] = fmaxf(p0.y, fmaxf(p1.y, p2.y)), hi[2] = fmaxf(p0.z, fmaxf(p1.z, p2.z));
        tri_box[2 * (size_t)g + 0] = make_float4(lo[0], lo[1], lo[2], 0.f);
        tri_box[2 * (size_t)g + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
        for (int k = 0; k < 3; ++k) slo[k] = fminf(slo[k], lo[k]), shi[k] = fmaxf(shi[k], hi[k]);
    }
    // wave reduction, workgroup reduction, then one atomic per workgroup and component
    for (int k = 0; k < 3; ++k)
    {
        float l = slo[k], h = shi[k];
        for (int off = 32; off > 0; off >>= 1)
        {
            l = fminf(l, __shfl_down(l, off));
            h = fmaxf(h, __shfl_down(h, off));
        }
        if ((threadIdx.x & 63u) == 0) s_lo[threadIdx.x >> 6][k] = l, s_hi[threadIdx.x >> 6][k] = h;
    }
    __syncthreads();
    if (threadIdx.x < 3u)
    {
        const int k = (int)threadIdx.x;
        float     l = s_lo[0][k], h = s_hi[0][k];
        for (uint32_t w = 1; w < kBlock / 64; ++w) l = fminf(l, s_lo[w][k]), h = fmaxf(h, s_hi[w][k]);
        if (l != INFINITY) atomicMin(&a.bounds[k], float_to_ordered(l));
        if (h != -INFINITY) atomicMax(&a.bounds[3 + k], float_to_ordered(h));
    }
}

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ __launch_bounds__(kBlock) void k_morton(BvhBuildArgs a)
{
    const float4* tri_box = a.tri_box;
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= a.tri_count) return;
    const float4 lo = tri_box[2 * (size_t)g], hi = tri_box[2 * (size_t)g + 1];
    uint32_t     q[3];
    const float  c[3] = {(lo.x + hi.x) * 0.5f, (lo.y + hi.y) * 0.5f, (lo.z + hi.z) * 0.5f};
    for (int k = 0; k < 3; ++k)
    {
        const float blo = ordered_to_float(a.bounds[k]), bhi = ordered_to_float(a.bounds[3 + k]);
        const float ext = bhi - blo;
        const float n   = ext > 0.0f ? (c[k] - blo) / ext : 0.0f;
        q[k]            = (uint32_t)fminf(fmaxf(n * 1024.0f, 0.0f), 1023.0f);
    }
    a.keys[0][g] = (expand_bits10(q[0]) << 2) | (expand_bits10(q[1]) << 1) | expand_bits10(q[2]);
    a.vals[0][g] = g;
}

// ---- stable LSD radix sort, 8 bits per pass ----
__global__ __launch_bounds__(kBlock) void k_radix_hist(const uint32_t* keys, uint32_t n, uint32_t shift, uint32_t nblocks, uint32_t* hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
    for (uint32_t k = threadIdx.x; k < kSortTile; k += kBlock)
    {
        const uint32_t i = base + k;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of hist (digit-major, block-minor) in three launches (round 6; one workgroup looping over 2 M entries with twenty
// barriers per 1024 took 3.6 ms per pass at 16.8 M triangles -- a third of the clustering build): per tile of kScanTile entries a sum,
// an exclusive scan of the tile sums by one workgroup, then every tile scans itself on top of its sum's prefix.
constexpr uint32_t kScanTile = 4096;  // 1024 threads x 4
__global__ __launch_bounds__(1024) void k_scan_tile_sums(const uint32_t* v, uint32_t total, uint32_t* sums)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t      i0 = blockIdx.x * kScanTile + threadIdx.x * 4u;
    uint32_t            s  = 0;
    for (uint32_t k = 0; k < 4u; ++k) s += i0 + k < total ? v[i0 + k] : 0u;
    for (int off = 32; off > 0; off >>= 1) s += (uint32_t)__shfl_xor((int)s, off);
    if ((threadIdx.x & 63u) == 0) s_wave[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t t = 0;
        for (int w = 0; w < 16; ++w) t += s_wave[w];
        sums[blockIdx.x] = t;
    }
}
// exclusive scan of n values by ONE workgroup (n <= a few thousand tile sums): per-thread runs + a Hillis-Steele scan of the run sums
__global__ __launch_bounds__(1024) void k_scan_small(uint32_t* v, uint32_t n)
{
    __shared__ uint32_t part[1024];
    const uint32_t      t = threadIdx.x, per = (n + 1023u) / 1024u, b0 = min(n, t * per), b1 = min(n, b0 + per);
    uint32_t            s = 0;
    for (uint32_t b = b0; b < b1; ++b) s += v[b];
    part[t] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1)
    {
        const uint32_t x = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    uint32_t e = part[t] - s;
    for (uint32_t b = b0; b < b1; ++b)
    {
        const uint32_t x = v[b];
        v[b] = e;
        e += x;
    }
}
__global__ __launch_bounds__(1024) void k_scan_tiles(uint32_t* v, uint32_t total, const uint32_t* sums)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t      i0 = blockIdx.x * kScanTile + threadIdx.x * 4u, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t            x[4], s = 0;
    for (uint32_t k = 0; k < 4u; ++k) x[k] = i0 + k < total ? v[i0 + k] : 0u, s += x[k];
    // inclusive scan of the threads' sums inside the wave, then across the 16 waves
    uint32_t incl = s;
    for (int off = 1; off < 64; off <<= 1)
    {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, off);
        if ((int)lane >= off) incl += y;
    }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = sums[blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w) base += s_wave[w];
    uint32_t e = base + incl - s;
    for (uint32_t k = 0; k < 4u; ++k)
    {
        if (i0 + k < total) v[i0 + k] = e;
        e += x[k];
    }
}
// scratch: ceil(total / kScanTile) words
static void launch_exclusive_scan(hipStream_t stream, uint32_t* v, uint32_t total, uint32_t* scratch)
{
    const uint32_t tiles = (total + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(1024), 0, stream, v, total, scratch);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, stream, scratch, tiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(1024), 0, stream, v, total, scratch);
}

__global__ __launch_bounds__(kBlock) void k_radix_scatter(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                                                          uint32_t* vals_out, uint32_t n, uint32_t shift, uint32_t nblocks,
                                                          const uint32_t* hist)
{
    __shared__ uint32_t base[256];
    __shared__ uint32_t wcount[kBlock / 64][256];
    const uint32_t      wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    base[threadIdx.x] = hist[threadIdx.x * nblocks + blockIdx.x];
    for (uint32_t w = 0; w < kBlock / 64; ++w) wcount[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile = blockIdx.x * kSortTile;
    for (uint32_t k = 0; k < kSortTile; k += kBlock)
    {
        const uint32_t i      = tile + k + threadIdx.x;
        const bool     active = i < n;
        uint32_t       key = 0, val = 0, digit = 0;
        if (active) key = keys_in[i], val = vals_in[i], digit = (key >> shift) & 255u;
        unsigned long long peers = __ballot(active);
        for (uint32_t b = 0; b < 8; ++b)
        {
            const unsigned long long m = __ballot(active && ((digit >> b) & 1u));
            peers &= ((digit >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (active && rank == 0) wcount[wave][digit] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (active)
        {
            uint32_t off = base[digit] + rank;
            for (uint32_t w = 0; w < wave; ++w) off += wcount[w][digit];
            keys_out[off] = key;
            vals_out[off] = val;
        }
        __syncthreads();
        uint32_t add = 0;
        for (uint32_t w = 0; w < kBlock / 64; ++w)
        {
            add += wcount[w][threadIdx.x];
            wcount[w][threadIdx.x] = 0;
        }
        base[threadIdx.x] += add;
        __syncthreads();
    }
}

// ---- Karras 2012 ----
__device__ __forceinline__ int delta(const uint32_t* keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = ((unsigned long long)keys[i] << 32) | (uint32_t)i, b = ((unsigned long long)keys[j] << 32) | (uint32_t)j;
    return __clzll((long long)(a ^ b));
}

__global__ __launch_bounds__(kBlock) void k_hierarchy(const uint32_t* keys, uint32_t n, float4* nodes, uint32_t* parent, uint32_t* subtree_count)
{
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
    const int N = (int)n;
    if (i >= N - 1) return;
    const int d    = (delta(keys, N, i, i + 1) - delta(keys, N, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, N, i, i - d);
    int       lmax = 2;
    while (delta(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
    const int j     = i + l * d;
    const int dnode = delta(keys, N, i, j);
    int       s = 0, t = l;
    do
    {
        t = (t + 1) >> 1;
        if (delta(keys, N, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = (lo == gamma) ? ~gamma : gamma, right = (hi == gamma + 1) ? ~(gamma + 1) : (gamma + 1);
    // q3 = (child0, child1, traversal child0, traversal child1); the boxes are filled by the refit.  The traversal pointers
    // treat a subtree of at most kLeafMax triangles as ONE leaf: its triangles are consecutive in sorted order, so the leaf is
    // the range code ~(first | (count - 1) << kLeafCountShift).  Fewer box tests and stack operations; the binary tree
    // (and its boxes) stays complete for the refit and for cap_bvh_readback.
    const int      nl = gamma - lo + 1, nr = hi - gamma;
    subtree_count[i] = (uint32_t)(hi - lo + 1);  // triangles below node i (the device collapse into the 8-wide view reads it)
    const uint32_t tleft  = nl <= kLeafMax ? ~((uint32_t)lo | ((uint32_t)(nl - 1) << kLeafCountShift)) : (uint32_t)gamma;
    const uint32_t tright = nr <= kLeafMax ? ~((uint32_t)(gamma + 1) | ((uint32_t)(nr - 1) << kLeafCountShift)) : (uint32_t)(gamma + 1);
    nodes[4 * (size_t)i + 3] = make_float4(u2f((uint32_t)left), u2f((uint32_t)right), u2f(tleft), u2f(tright));
    const uint32_t pl = ((uint32_t)i << 1), pr = ((uint32_t)i << 1) | 1u;
    if (left < 0) parent[(N - 1) + gamma] = pl; else parent[gamma] = pl;
    if (right < 0) parent[(N - 1) + gamma + 1] = pr; else parent[gamma + 1] = pr;
    if (i == 0) parent[0] = 0xffffffffu;
}

__device__ __forceinline__ float load_agent(const float* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One thread per leaf: copy the triangle into leaf order, then climb.  The first thread to reach a node parks its
// box in the node and leaves; the second merges both boxes and continues, so every internal node is completed once.
__global__ __launch_bounds__(kBlock) void k_refit(BvhBuildArgs a, const uint32_t* vals_sorted)
{
    const float4* tri_box = a.tri_box;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t n = a.tri_count;
    if (i >= n) return;
    const uint32_t g = vals_sorted[i];
    a.leaf_tri[i]    = g;
    a.tris_sorted[4 * (size_t)i + 0] = a.tri_raw[4 * (size_t)g + 0];
    a.tris_sorted[4 * (size_t)i + 1] = a.tri_raw[4 * (size_t)g + 1];
    a.tris_sorted[4 * (size_t)i + 2] = a.tri_raw[4 * (size_t)g + 2];
    a.tris_sorted[4 * (size_t)i + 3] = a.tri_raw[4 * (size_t)g + 3];
    if (n < 2) return;
    const float4 blo = tri_box[2 * (size_t)g], bhi = tri_box[2 * (size_t)g + 1];
    float        lo[3] = {blo.x, blo.y, blo.z}, hi[3] = {bhi.x, bhi.y, bhi.z};
    // pad: the box must contain every point the fp32 triangle test can report as a hit
    for (int k = 0; k < 3; ++k)
    {
        const float pad = 1e-5f * fmaxf(1.0f, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
        lo[k] -= pad, hi[k] += pad;
    }
    uint32_t cur = a.parent[(n - 1) + i];
    while (cur != 0xffffffffu)
    {
        const uint32_t p = cur >> 1, slot = cur & 1u;
        float*         q = reinterpret_cast<float*>(a.nodes + 4 * (size_t)p);
        float*         mine  = q + (slot ? 6 : 0);
        const float*   other = q + (slot ? 0 : 6);
        mine[0] = lo[0], mine[1] = lo[1], mine[2] = lo[2], mine[3] = hi[0], mine[4] = hi[1], mine[5] = hi[2];
        __threadfence();
        const uint32_t arrived = __hip_atomic_fetch_add(&a.flags[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == 0) return;
        __threadfence();
        for (int k = 0; k < 3; ++k)
        {
            lo[k] = fminf(lo[k], load_agent(other + k));
            hi[k] = fmaxf(hi[k], load_agent(other + 3 + k));
        }
        cur = a.parent[p];
    }
}

__global__ __launch_bounds__(kBlock) void k_depth(const uint32_t* parent, uint32_t n, uint32_t* max_depth)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t       depth = 0;
    if (i < n && n >= 2)
    {
        uint32_t cur = parent[(n - 1) + i];
        while (cur != 0xffffffffu)
        {
            ++depth;
            cur = parent[cur >> 1];
        }
    }
    for (int off = 32; off > 0; off >>= 1) depth = max(depth, (uint32_t)__shfl_down(depth, off));
    if ((threadIdx.x & 63u) == 0 && depth) atomicMax(max_depth, depth);
}
// Host-built tree (sah_builder.cpp): copy the intersection records into leaf order.
__global__ __launch_bounds__(kBlock) void k_gather_sorted(BvhBuildArgs a)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.tri_count) return;
    const uint32_t g = a.leaf_tri[i];
    for (int k = 0; k < 4; ++k) a.tris_sorted[4 * (size_t)i + k] = a.tri_raw[4 * (size_t)g + k];
}
// Intersection records in the leaf order of the compressed 8-wide view (cap_wide.h): record i = leaf-order record tri_src[i].
__global__ __launch_bounds__(kBlock) void k_gather_wide(const uint32_t* tri_src, const float4* tris_sorted, uint32_t n, float4* tris8)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = tri_src[i];
    for (int k = 0; k < 4; ++k) tris8[4 * (size_t)i + k] = tris_sorted[4 * (size_t)s + k];
}
}  // namespace

void launch_gather_wide(hipStream_t stream, const uint32_t* tri_src, const float4* tris_sorted, uint32_t n, float4* tris8)
{
    if (n) hipLaunchKernelGGL(k_gather_wide, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, tri_src, tris_sorted, n, tris8);
}

// ------------------------------------------------------------------------------------------------
// Device-side collapse of the binary tree into the compressed 8-wide view (cap_wide.h): the same greedy rules as the host
// builder (wide_builder.cpp: open the largest inner child until eight, then multi-triangle leaf children; octant slots by greedy
// assignment; planes quantised outwards in double precision after padding), one thread per wide node, level by level.  A level's
// nodes are contiguous (breadth-first array, top levels first); a node's inner children are allocated as one block, so they
// are contiguous in slot order, and its triangles likewise.  The blocks are handed out by a prefix sum in the level's own order (round 6;
// an atomic per node before): the same layout on every run and on every rank, siblings' children next to each other.
// ------------------------------------------------------------------------------------------------
namespace
{
struct WideRef
{
    int      node;   // >= 0: binary internal node, < 0: ~(leaf-order triangle)
    uint32_t count;  // triangles below
    float    lo[3], hi[3];
};
__device__ __forceinline__ WideRef wide_ref(const float4* bn, const uint32_t* count, uint32_t node, int s)
{
    const float* q = reinterpret_cast<const float*>(bn + 4 * (size_t)node);
    WideRef      r;
    for (int k = 0; k < 3; ++k) r.lo[k] = q[6 * s + k], r.hi[k] = q[6 * s + 3 + k];
    r.node  = (int)f2u(q[12 + s]);
    r.count = r.node >= 0 ? count[r.node] : 1u;
    return r;
}
__device__ __forceinline__ double wide_half_area(const WideRef& r)
{
    const double dx = (double)r.hi[0] - r.lo[0], dy = (double)r.hi[1] - r.lo[1], dz = (double)r.hi[2] - r.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// the children of the wide node that stands for binary node `root`: open the largest inner child until eight, then multi-triangle leaves
__device__ __forceinline__ int wide_open(const WideCollapseArgs& a, uint32_t root, WideRef kids[8])
{
    int nk = 2;
    kids[0] = wide_ref(a.bnodes, a.count, root, 0), kids[1] = wide_ref(a.bnodes, a.count, root, 1);
    for (int pass = 0; pass < 2; ++pass)
        while (nk < 8)
        {
            int    best = -1;
            double best_area = -1.0;
            for (int i = 0; i < nk; ++i)
            {
                const bool inner = kids[i].count > kWideLeafMax;
                const bool can   = pass == 0 ? inner : (!inner && kids[i].count > 1u);
                const double ar  = wide_half_area(kids[i]);
                if (can && ar > best_area) best_area = ar, best = i;
            }
            if (best < 0) break;
            const uint32_t open = (uint32_t)kids[best].node;
            kids[best]  = wide_ref(a.bnodes, a.count, open, 0);
            kids[nk++]  = wide_ref(a.bnodes, a.count, open, 1);
        }
    return nk;
}

// first half of a level: how many inner children and triangles every node of the level will allocate
__global__ __launch_bounds__(kBlock) void k_wide_count(WideCollapseArgs a)
{
    const uint32_t w = a.begin + blockIdx.x * kBlock + threadIdx.x;
    if (w >= a.end) return;
    WideRef   kids[8];
    const int nk = wide_open(a, a.task[w], kids);
    uint32_t  n_inner = 0u, n_tris = 0u;
    for (int i = 0; i < nk; ++i)
        if (kids[i].count > kWideLeafMax)
            ++n_inner;
        else
            n_tris += kids[i].count;
    a.cnt[2 * (size_t)(w - a.begin)] = n_inner, a.cnt[2 * (size_t)(w - a.begin) + 1] = n_tris;
}

// exclusive scan of the level's (inner children, triangles) pairs on top of alloc[0..1]; alloc takes the totals.  The blocks a level's
// nodes get are in the level's own order: the layout is the same on every run and a node's children lie next to its siblings' children.
// Three launches (tile sums, their scan by one workgroup, the tiles): a level of the 16.8 M-triangle tree has 1.4 M nodes.
constexpr uint32_t kWideScanTile = 1024;  // pairs per workgroup (256 threads x 4)
__global__ __launch_bounds__(kBlock) void k_wide_scan_sums(WideCollapseArgs a, uint32_t* sums)
{
    __shared__ uint32_t s_wave[kBlock / 64][2];
    const uint32_t      n = a.end - a.begin, i0 = blockIdx.x * kWideScanTile + threadIdx.x * 4u;
    uint32_t            s0 = 0, s1 = 0;
    for (uint32_t k = 0; k < 4u; ++k)
        if (i0 + k < n) s0 += a.cnt[2 * (size_t)(i0 + k)], s1 += a.cnt[2 * (size_t)(i0 + k) + 1];
    for (int off = 32; off > 0; off >>= 1) s0 += (uint32_t)__shfl_xor((int)s0, off), s1 += (uint32_t)__shfl_xor((int)s1, off);
    if ((threadIdx.x & 63u) == 0) s_wave[threadIdx.x >> 6][0] = s0, s_wave[threadIdx.x >> 6][1] = s1;
    __syncthreads();
    if (threadIdx.x < 2u)
    {
        uint32_t t = 0;
        for (uint32_t w = 0; w < kBlock / 64; ++w) t += s_wave[w][threadIdx.x];
        sums[2 * (size_t)blockIdx.x + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void k_wide_scan_small(WideCollapseArgs a, uint32_t* sums, uint32_t tiles)
{
    __shared__ uint32_t s_part[2][1024];
    const uint32_t t = threadIdx.x, per = (tiles + 1023u) / 1024u, b0 = min(tiles, t * per), b1 = min(tiles, b0 + per);
    uint32_t       s0 = 0, s1 = 0;
    for (uint32_t b = b0; b < b1; ++b) s0 += sums[2 * (size_t)b], s1 += sums[2 * (size_t)b + 1];
    s_part[0][t] = s0, s_part[1][t] = s1;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1)
    {
        const uint32_t v0 = t >= off ? s_part[0][t - off] : 0u, v1 = t >= off ? s_part[1][t - off] : 0u;
        __syncthreads();
        s_part[0][t] += v0, s_part[1][t] += v1;
        __syncthreads();
    }
    uint32_t e0 = a.alloc[0] + s_part[0][t] - s0, e1 = a.alloc[1] + s_part[1][t] - s1;
    for (uint32_t b = b0; b < b1; ++b)
    {
        const uint32_t k0 = sums[2 * (size_t)b], k1 = sums[2 * (size_t)b + 1];
        sums[2 * (size_t)b] = e0, sums[2 * (size_t)b + 1] = e1;
        e0 += k0, e1 += k1;
    }
    __syncthreads();  // every read of alloc is done
    if (t == 1023u) a.alloc[0] += s_part[0][t], a.alloc[1] += s_part[1][t];
}
__global__ __launch_bounds__(kBlock) void k_wide_scan_tiles(WideCollapseArgs a, const uint32_t* sums)
{
    __shared__ uint32_t s_wave[kBlock / 64][2];
    const uint32_t      n = a.end - a.begin, i0 = blockIdx.x * kWideScanTile + threadIdx.x * 4u, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t            x0[4], x1[4], s0 = 0, s1 = 0;
    for (uint32_t k = 0; k < 4u; ++k)
    {
        x0[k] = i0 + k < n ? a.cnt[2 * (size_t)(i0 + k)] : 0u, x1[k] = i0 + k < n ? a.cnt[2 * (size_t)(i0 + k) + 1] : 0u;
        s0 += x0[k], s1 += x1[k];
    }
    uint32_t c0 = s0, c1 = s1;
    for (int off = 1; off < 64; off <<= 1)
    {
        const uint32_t y0 = (uint32_t)__shfl_up((int)c0, off), y1 = (uint32_t)__shfl_up((int)c1, off);
        if ((int)lane >= off) c0 += y0, c1 += y1;
    }
    if (lane == 63u) s_wave[wave][0] = c0, s_wave[wave][1] = c1;
    __syncthreads();
    uint32_t e0 = sums[2 * (size_t)blockIdx.x] + c0 - s0, e1 = sums[2 * (size_t)blockIdx.x + 1] + c1 - s1;
    for (uint32_t w = 0; w < wave; ++w) e0 += s_wave[w][0], e1 += s_wave[w][1];
    for (uint32_t k = 0; k < 4u; ++k)
    {
        if (i0 + k < n) a.cnt[2 * (size_t)(i0 + k)] = e0, a.cnt[2 * (size_t)(i0 + k) + 1] = e1;
        e0 += x0[k], e1 += x1[k];
    }
}

__global__ __launch_bounds__(kBlock) void k_wide_level(WideCollapseArgs a)
{
    const uint32_t w = a.begin + blockIdx.x * kBlock + threadIdx.x;
    if (w >= a.end) return;
    WideRef   kids[8];
    const int nk = wide_open(a, a.task[w], kids);
    // padded child boxes (double) and the node box
    double clo[8][3], chi[8][3], nlo[3] = {1e300, 1e300, 1e300}, nhi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < nk; ++i)
        for (int k = 0; k < 3; ++k)
        {
            clo[i][k] = (double)kids[i].lo[k] - a.pad, chi[i][k] = (double)kids[i].hi[k] + a.pad;
            nlo[k] = fmin(nlo[k], clo[i][k]), nhi[k] = fmax(nhi[k], chi[i][k]);
        }
    // slot assignment: greedy maximum of <child centre - node centre, direction(slot)>
    int  kid_at[8];
    bool slot_used[8], kid_done[8];
    for (int s = 0; s < 8; ++s) kid_at[s] = -1, slot_used[s] = false, kid_done[s] = false;
    for (int step = 0; step < nk; ++step)
    {
        double best = -1e300;
        int    bi = -1, bs = -1;
        for (int i = 0; i < nk; ++i)
        {
            if (kid_done[i]) continue;
            double off[3];
            for (int k = 0; k < 3; ++k) off[k] = 0.5 * (clo[i][k] + chi[i][k]) - 0.5 * (nlo[k] + nhi[k]);
            for (int s = 0; s < 8; ++s)
            {
                if (slot_used[s]) continue;
                const double v = ((s & 1) ? off[0] : -off[0]) + ((s & 2) ? off[1] : -off[1]) + ((s & 4) ? off[2] : -off[2]);
                if (v > best) best = v, bi = i, bs = s;
            }
        }
        kid_at[bs] = bi, slot_used[bs] = true, kid_done[bi] = true;
    }
    uint32_t word[kWideNodeWords];
    for (uint32_t k = 0; k < kWideNodeWords; ++k) word[k] = 0u;
    // grid origin (the node's low corner rounded down to float) and steps (the smallest power of two with <= 255 steps)
    float  p[3];
    double step[3];
    uint32_t eb[3];
    for (int k = 0; k < 3; ++k)
    {
        p[k] = (float)nlo[k];
        if ((double)p[k] > nlo[k]) p[k] = u2f(p[k] > 0.0f ? f2u(p[k]) - 1u : (p[k] < 0.0f ? f2u(p[k]) + 1u : 0x80000001u));  // next float down
        word[k] = f2u(p[k]);
        const double ext = nhi[k] - (double)p[k];
        int          e   = -100;
        if (ext > 0.0)
        {
            int fe;
            (void)frexp(ext / 255.0, &fe);
            e = fe - 1 > -100 ? fe - 1 : -100;
        }
        while (ceil(ext / ldexp(1.0, e)) > 255.0) ++e;
        step[k] = ldexp(1.0, e);
        eb[k]   = (uint32_t)(e + 127);
    }
    word[3] = eb[0] << 23;
    word[7] = ((eb[1] << 23) & 0xffff0000u) | ((eb[2] << 23) >> 16);
    uint32_t imask = 0u, tvalid = 0u, n_inner = 0u, n_tris = 0u;
    uint32_t leaf_tri[8][kWideLeafMax];
    uint32_t leaf_n[8];
    for (int s = 0; s < 8; ++s)
    {
        leaf_n[s] = 0u;
        const int i = kid_at[s];
        if (i < 0) continue;
        if (kids[i].count > kWideLeafMax)
            imask |= 1u << s, ++n_inner;
        else
        {
            // the (at most kWideLeafMax) triangles below: a tiny explicit stack over the binary subtree
            int st[4], sp = 0;
            st[sp++] = kids[i].node;
            while (sp > 0)
            {
                const int c = st[--sp];
                if (c < 0)
                    leaf_tri[s][leaf_n[s]++] = (uint32_t)~c;
                else
                {
                    const float* q = reinterpret_cast<const float*>(a.bnodes + 4 * (size_t)c);
                    st[sp++] = (int)f2u(q[13]), st[sp++] = (int)f2u(q[12]);
                }
            }
            for (uint32_t k = 0; k < leaf_n[s]; ++k) tvalid |= 1u << (k * 8u + (uint32_t)s);
            n_tris += leaf_n[s];
        }
        for (int k = 0; k < 3; ++k)
        {
            double qlo = floor((clo[i][k] - (double)p[k]) / step[k]), qhi = ceil((chi[i][k] - (double)p[k]) / step[k]);
            qlo = fmin(fmax(qlo, 0.0), 255.0), qhi = fmin(fmax(qhi, 0.0), 255.0);
            const uint32_t wi = 8u + 2u * (uint32_t)k + ((uint32_t)s >> 2), sh = 8u * ((uint32_t)s & 3u);
            word[wi] |= (uint32_t)qlo << sh;
            word[wi + 6] |= (uint32_t)qhi << sh;
        }
    }
    const uint32_t child_base = n_inner ? a.cnt[2 * (size_t)(w - a.begin)] : 0u;
    const uint32_t tri_base   = n_tris ? a.cnt[2 * (size_t)(w - a.begin) + 1] : 0u;
    word[4] = child_base, word[5] = tri_base, word[6] = tvalid | (imask << 24);
    uint32_t rel = 0u, at = tri_base;
    for (int s = 0; s < 8; ++s)
        if (imask & (1u << s)) a.task[child_base + rel++] = (uint32_t)kids[kid_at[s]].node;
    for (uint32_t k = 0; k < kWideLeafMax; ++k)
        for (int s = 0; s < 8; ++s)
            if (leaf_n[s] > k) a.tri_src[at++] = leaf_tri[s][k];
    uint32_t* o = a.nodes8 + (size_t)w * kWideNodeStride;
    for (uint32_t k = 0; k < kWideNodeWords; ++k) o[k] = word[k];
}
}  // namespace

int launch_wide_collapse(hipStream_t stream, WideCollapseArgs a, uint32_t* node_count, uint32_t* depth, uint32_t* top_nodes)
{
    // level 0 = the root (binary node 0); alloc[0] counts allocated wide nodes, alloc[1] emitted triangles
    const uint32_t init[2] = {1u, 0u}, root_task = 0u;
    if (hipMemcpyAsync(a.alloc, init, sizeof(init), hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    if (hipMemcpyAsync(a.task, &root_task, sizeof(root_task), hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    uint32_t begin = 0u, end = 1u, levels = 0u, top = 0u;
    while (begin < end)
    {
        ++levels;
        if (levels <= 3) top = end < kWideTopNodes ? end : kWideTopNodes;
        a.begin = begin, a.end = end;
        hipLaunchKernelGGL(k_wide_count, dim3((end - begin + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, a);
        {
            const uint32_t tiles = (end - begin + kWideScanTile - 1) / kWideScanTile;
            uint32_t*      sums  = a.cnt + 2 * (size_t)a.capacity;  // (the tail of the per-level scratch: 2 * (capacity / 1024 + 2) words)
            hipLaunchKernelGGL(k_wide_scan_sums, dim3(tiles), dim3(kBlock), 0, stream, a, sums);
            hipLaunchKernelGGL(k_wide_scan_small, dim3(1), dim3(1024), 0, stream, a, sums, tiles);
            hipLaunchKernelGGL(k_wide_scan_tiles, dim3(tiles), dim3(kBlock), 0, stream, a, sums);
        }
        hipLaunchKernelGGL(k_wide_level, dim3((end - begin + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, a);
        uint32_t allocated = 0u;
        if (hipMemcpyAsync(&allocated, a.alloc, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
        if (hipStreamSynchronize(stream) != hipSuccess) return 1;
        if (allocated > a.capacity) return 2;  // cannot happen: a wide node stands for >= 4 triangles
        begin = end, end = allocated;
    }
    *node_count = end, *depth = levels, *top_nodes = top;
    return 0;
}

size_t bvh_radix_blocks(uint32_t n) { return (n + kSortTile - 1) / kSortTile; }

static void bvh_setup(hipStream_t stream, const BvhBuildArgs& a)
{
    const uint32_t n      = a.tri_count;
    const uint32_t blocks = (n + kBlock - 1) / kBlock;
    // bounds = (+inf, +inf, +inf, -inf, -inf, -inf) in the ordered encoding; flags, depth = 0
    const uint32_t init[6] = {0xFF800000u, 0xFF800000u, 0xFF800000u, 0x007FFFFFu, 0x007FFFFFu, 0x007FFFFFu};
    (void)hipMemcpyAsync(a.bounds, init, sizeof(init), hipMemcpyHostToDevice, stream);
    (void)hipMemsetAsync(a.flags, 0, sizeof(uint32_t) * n, stream);
    (void)hipMemsetAsync(a.max_depth, 0, sizeof(uint32_t), stream);
    hipLaunchKernelGGL(k_tri_setup, dim3(blocks < 4096u ? blocks : 4096u), dim3(kBlock), 0, stream, a);
}

void launch_bvh_setup(hipStream_t stream, const BvhBuildArgs& a)
{
    if (a.tri_count) bvh_setup(stream, a);
}

void launch_bvh_finish_host(hipStream_t stream, const BvhBuildArgs& a)
{
    const uint32_t n = a.tri_count;
    if (n == 0) return;
    const uint32_t blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_gather_sorted, dim3(blocks), dim3(kBlock), 0, stream, a);
}


// setup + Morton codes + radix sort; the sorted (key, triangle) arrays are a.keys[r], a.vals[r] for the returned r
int launch_bvh_sort(hipStream_t stream, const BvhBuildArgs& a)
{
    const uint32_t n = a.tri_count;
    if (n == 0) return 0;
    const uint32_t blocks = (n + kBlock - 1) / kBlock;
    bvh_setup(stream, a);
    int src = 0;
    if (n >= 2)
    {
        hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(kBlock), 0, stream, a);
        const uint32_t nb = (uint32_t)bvh_radix_blocks(n);
        for (uint32_t shift = 0; shift < 32; shift += 8)
        {
            hipLaunchKernelGGL(k_radix_hist, dim3(nb), dim3(kBlock), 0, stream, a.keys[src], n, shift, nb, a.hist);
            launch_exclusive_scan(stream, a.hist, 256u * nb, a.parent);  // (a.parent: 2 n words that every builder writes in full after the sort)
            hipLaunchKernelGGL(k_radix_scatter, dim3(nb), dim3(kBlock), 0, stream, a.keys[src], a.vals[src], a.keys[src ^ 1],
                               a.vals[src ^ 1], n, shift, nb, a.hist);
            src ^= 1;
        }
    }
    else
    {
        const uint32_t zero = 0;
        (void)hipMemcpyAsync(a.vals[0], &zero, sizeof(zero), hipMemcpyHostToDevice, stream);
    }
    return src;
}

void launch_bvh_build(hipStream_t stream, const BvhBuildArgs& a)
{
    const uint32_t n = a.tri_count;
    if (n == 0) return;
    const uint32_t blocks = (n + kBlock - 1) / kBlock;
    const int      src    = launch_bvh_sort(stream, a);
    if (n >= 2)
        hipLaunchKernelGGL(k_hierarchy, dim3(blocks), dim3(kBlock), 0, stream, a.keys[src], n, a.nodes, a.parent, a.keys[src ^ 1]);  // the other key buffer is free after the sort
    hipLaunchKernelGGL(k_refit, dim3(blocks), dim3(kBlock), 0, stream, a, a.vals[src]);
    hipLaunchKernelGGL(k_depth, dim3(blocks), dim3(kBlock), 0, stream, a.parent, n, a.max_depth);
}
}  // namespace cap
