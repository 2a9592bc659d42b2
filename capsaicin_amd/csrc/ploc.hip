// ploc.hip — on-device agglomerative tree build for gfx950: parallel locally-ordered clustering (the published algorithm of
// Meister & Bittner, "Parallel Locally-Ordered Clustering for Bounding Volume Hierarchy Construction", 2018, restated for
// 64-wide waves).  Same role as bvh.hip's Morton hierarchy -- it replaces the driver's acceleration-structure build the
// reference asks for with PREFER_FAST_TRACE (src/systems/blas_system.cpp:42-65) -- with the tree quality of a surface-area
// build and no host round trip of the geometry.
//
//   clusters = the triangles in Morton order (bvh.hip's setup + sort)
//   repeat:  nn[i]   = the cluster within `radius` array positions whose union with i has the smallest surface area
//            i and nn[i] merge into a new node when nn[nn[i]] == i;  the node takes the lower position
//            compact the array (order kept: it stays a space-filling-curve order)
//   until one cluster is left.
//
// Every step is deterministic (ties by a symmetric pair key, node numbers from a prefix sum), so the tree is the same on every
// run.  Node numbers count down from n - 2: the last merge is node 0, the root all traversals start from.  While more than
// kPlocTail clusters are left an iteration is four launches (search, count, scan, merge) and one 4-byte read-back; the rest
// runs in one workgroup out of LDS.  Two passes then put the leaves into depth-first order (a subtree's triangles
// consecutive: the traversal leaves ~(first | count - 1 << kLeafCountShift) of cap_leaf.h need that) and rewrite the links.
#include "cap_kernels.h"

namespace cap
{
namespace
{
constexpr uint32_t kPlocBlock     = 256;
constexpr uint32_t kPlocMaxRadius = 32;
constexpr uint32_t kPlocTail      = 1024;

struct PlocArgs
{
    uint32_t        n, radius;
    const float4*   tri_box;      // 2 per triangle, global order
    const uint32_t* order;        // Morton rank -> global triangle
    const float4*   tri_raw;
    float4 *        lo[2], *hi[2];  // cluster boxes; lo.w = triangles below, hi.w = height (0 for a triangle)
    int*            ref[2];         // >= 0: node, < 0: ~(Morton rank)
    int*            nn;
    int*            child;          // 2 per node: the merged clusters' refs (ranks, not yet leaf positions)
    uint32_t*       blk;            // 2 per workgroup: kept clusters, new nodes
    uint32_t*       ctrl;           // [2 * parity]: clusters, nodes created so far
    float4*         nodes;
    uint32_t*       parent;         // as bvh.hip: nodes, then leaves by Morton rank; (parent << 1) | slot
    uint32_t*       count;          // triangles below node
    uint32_t*       leaf_pos;       // Morton rank -> depth-first position
    uint32_t*       leaf_tri;
    float4*         tris_sorted;
    uint32_t*       max_depth;
};

__device__ __forceinline__ float union_half_area(const float4& alo, const float4& ahi, const float4& blo, const float4& bhi)
{
    const float dx = fmaxf(ahi.x, bhi.x) - fminf(alo.x, blo.x), dy = fmaxf(ahi.y, bhi.y) - fminf(alo.y, blo.y),
                dz = fmaxf(ahi.z, bhi.z) - fminf(alo.z, blo.z);
    return dx * dy + dy * dz + dz * dx;
}

__global__ __launch_bounds__(kPlocBlock) void k_ploc_init(PlocArgs a)
{
    const uint32_t i = blockIdx.x * kPlocBlock + threadIdx.x;
    if (i == 0) a.ctrl[0] = a.n, a.ctrl[1] = 0u;
    if (i >= a.n) return;
    const uint32_t g = a.order[i];
    float4         lo = a.tri_box[2 * (size_t)g], hi = a.tri_box[2 * (size_t)g + 1];
    // the leaf pad of bvh.hip's refit: the box must contain every point the fp32 triangle test can report as a hit
    float* l = &lo.x;
    float* h = &hi.x;
    for (int k = 0; k < 3; ++k)
    {
        const float pad = 1e-5f * fmaxf(1.0f, fmaxf(fabsf(l[k]), fabsf(h[k])));
        l[k] -= pad, h[k] += pad;
    }
    lo.w = u2f(1u), hi.w = u2f(0u);
    a.lo[0][i] = lo, a.hi[0][i] = hi, a.ref[0][i] = ~(int)i;
}

// Equal distances (copies of one triangle, regular grids) are ordered by a key that is symmetric in the pair -- closer positions
// first, then pairs whose lower position is even, then the lower position -- so that both ends of a pair agree on it: with every
// distance equal, positions (0, 1), (2, 3), ... are mutual nearest neighbours and the array halves per iteration (an asymmetric
// rule such as "the lower position wins" merges one pair per iteration there and builds a chain of depth n).
__device__ __forceinline__ uint32_t ploc_tie_key(int i, int j)
{
    const uint32_t a = (uint32_t)(i < j ? i : j), d = (uint32_t)(i < j ? j - i : i - j);
    return (d << 26) | ((a & 1u) << 25) | (a & 0x1ffffffu);  // d <= kPlocMaxRadius; the last term only orders equal (d, parity)
}

// the nearest neighbour of cluster `i` among positions [i - radius, i + radius] of an array of m; tile[] holds the boxes of
// positions tile_first ...
template <typename Box>
__device__ __forceinline__ int ploc_nearest(const Box* t_lo, const Box* t_hi, int tile_first, int i, int m, int radius)
{
    const float4 lo = t_lo[i - tile_first], hi = t_hi[i - tile_first];
    float        best = INFINITY;
    int          bj   = -1;
    const int    j0 = i - radius < 0 ? 0 : i - radius, j1 = i + radius > m - 1 ? m - 1 : i + radius;
    for (int j = j0; j <= j1; ++j)
    {
        if (j == i) continue;
        const float ar = union_half_area(lo, hi, t_lo[j - tile_first], t_hi[j - tile_first]);
        if (ar < best || (ar == best && ploc_tie_key(i, j) < ploc_tie_key(i, bj))) best = ar, bj = j;
    }
    if (bj < 0) bj = i > 0 ? i - 1 : i + 1;  // only if every area is inf / nan
    return bj;
}

__global__ __launch_bounds__(kPlocBlock) void k_ploc_nn(PlocArgs a, uint32_t p)
{
    __shared__ float4 s_lo[kPlocBlock + 2 * kPlocMaxRadius], s_hi[kPlocBlock + 2 * kPlocMaxRadius];
    const int m = (int)a.ctrl[2 * p], base = (int)(blockIdx.x * kPlocBlock), R = (int)a.radius;
    if (base >= m) return;
    for (int t = (int)threadIdx.x; t < (int)kPlocBlock + 2 * R; t += (int)kPlocBlock)
    {
        const int j = base - R + t;
        if (j >= 0 && j < m) s_lo[t] = a.lo[p][j], s_hi[t] = a.hi[p][j];
    }
    __syncthreads();
    const int i = base + (int)threadIdx.x;
    if (i < m) a.nn[i] = ploc_nearest(s_lo, s_hi, base - R, i, m, R);
}

// (keeps its position or becomes a node, becomes a node) of cluster i
__device__ __forceinline__ void ploc_flags(const int* nn, int i, int m, bool& keep, bool& create, int& j)
{
    keep = create = false, j = 0;
    if (i >= m) return;
    j                 = nn[i];
    const bool mutual = nn[j] == i;
    create            = mutual && i < j;
    keep              = !(mutual && i > j);
}

__global__ __launch_bounds__(kPlocBlock) void k_ploc_count(PlocArgs a, uint32_t p)
{
    __shared__ uint32_t s_sum[2];
    const int m = (int)a.ctrl[2 * p], i = (int)(blockIdx.x * kPlocBlock + threadIdx.x);
    if (threadIdx.x < 2) s_sum[threadIdx.x] = 0u;
    __syncthreads();
    bool keep, create;
    int  j;
    ploc_flags(a.nn, i, m, keep, create, j);
    const uint32_t nk = (uint32_t)__popcll(__ballot(keep)), nc = (uint32_t)__popcll(__ballot(create));
    if ((threadIdx.x & 63u) == 0) atomicAdd(&s_sum[0], nk), atomicAdd(&s_sum[1], nc);
    __syncthreads();
    if (threadIdx.x < 2) a.blk[2 * blockIdx.x + threadIdx.x] = s_sum[threadIdx.x];
}

// exclusive scan of the per-workgroup (kept, created) pairs, one workgroup; the totals become the next iteration's state
__global__ __launch_bounds__(1024) void k_ploc_scan(PlocArgs a, uint32_t p)
{
    __shared__ uint32_t s_part[2][1024];
    const uint32_t m = a.ctrl[2 * p], nb = (m + kPlocBlock - 1) / kPlocBlock, t = threadIdx.x;
    const uint32_t per = (nb + 1023u) / 1024u, b0 = t * per, b1 = min(nb, b0 + per);
    uint32_t       sk = 0, sc = 0;
    for (uint32_t b = b0; b < b1; ++b) sk += a.blk[2 * b], sc += a.blk[2 * b + 1];
    s_part[0][t] = sk, s_part[1][t] = sc;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1)
    {
        const uint32_t vk = t >= off ? s_part[0][t - off] : 0u, vc = t >= off ? s_part[1][t - off] : 0u;
        __syncthreads();
        s_part[0][t] += vk, s_part[1][t] += vc;
        __syncthreads();
    }
    uint32_t ek = s_part[0][t] - sk, ec = s_part[1][t] - sc;  // exclusive prefix of this thread's run of workgroups
    for (uint32_t b = b0; b < b1; ++b)
    {
        const uint32_t k = a.blk[2 * b], c = a.blk[2 * b + 1];
        a.blk[2 * b] = ek, a.blk[2 * b + 1] = ec;
        ek += k, ec += c;
    }
    if (t == 1023u) a.ctrl[2 * (p ^ 1u)] = s_part[0][t], a.ctrl[2 * (p ^ 1u) + 1] = a.ctrl[2 * p + 1] + s_part[1][t];
}

// the node (A, B) -> `node`, A the cluster of the lower position; returns the merged cluster
__device__ __forceinline__ void ploc_emit(const PlocArgs& a, uint32_t node, const float4& alo, const float4& ahi, int aref,
                                          const float4& blo, const float4& bhi, int bref, float4& lo, float4& hi)
{
    float4* q = a.nodes + 4 * (size_t)node;
    q[0]      = make_float4(alo.x, alo.y, alo.z, ahi.x);
    q[1]      = make_float4(ahi.y, ahi.z, blo.x, blo.y);
    q[2]      = make_float4(blo.z, bhi.x, bhi.y, bhi.z);
    a.child[2 * (size_t)node] = aref, a.child[2 * (size_t)node + 1] = bref;
    const uint32_t cnt = f2u(alo.w) + f2u(blo.w), height = max(f2u(ahi.w), f2u(bhi.w)) + 1u;
    a.count[node]      = cnt;
    a.parent[aref >= 0 ? (size_t)aref : (size_t)(a.n - 1u) + (size_t)~aref] = node << 1;
    a.parent[bref >= 0 ? (size_t)bref : (size_t)(a.n - 1u) + (size_t)~bref] = (node << 1) | 1u;
    lo = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), u2f(cnt));
    hi = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), u2f(height));
}

__global__ __launch_bounds__(kPlocBlock) void k_ploc_merge(PlocArgs a, uint32_t p)
{
    __shared__ uint32_t s_wave[2][kPlocBlock / 64];
    const int m = (int)a.ctrl[2 * p], i = (int)(blockIdx.x * kPlocBlock + threadIdx.x);
    const uint32_t created = a.ctrl[2 * p + 1], q = p ^ 1u;
    bool keep, create;
    int  j;
    ploc_flags(a.nn, i, m, keep, create, j);
    const unsigned long long bk = __ballot(keep), bc = __ballot(create);
    const uint32_t           lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long below = lane ? (~0ull >> (64u - lane)) : 0ull;
    if (lane == 0) s_wave[0][wave] = (uint32_t)__popcll(bk), s_wave[1][wave] = (uint32_t)__popcll(bc);
    __syncthreads();
    uint32_t pk = a.blk[2 * blockIdx.x] + (uint32_t)__popcll(bk & below), pc = a.blk[2 * blockIdx.x + 1] + (uint32_t)__popcll(bc & below);
    for (uint32_t w = 0; w < wave; ++w) pk += s_wave[0][w], pc += s_wave[1][w];
    if (!keep) return;
    float4 lo = a.lo[p][i], hi = a.hi[p][i];
    int    ref = a.ref[p][i];
    if (create)
    {
        const uint32_t node = (a.n - 2u) - (created + pc);
        float4         mlo, mhi;
        ploc_emit(a, node, lo, hi, ref, a.lo[p][j], a.hi[p][j], a.ref[p][j], mlo, mhi);
        lo = mlo, hi = mhi, ref = (int)node;
    }
    a.lo[q][pk] = lo, a.hi[q][pk] = hi, a.ref[q][pk] = ref;
}

// the last <= kPlocTail clusters: one workgroup, the cluster array in LDS, no launches between iterations
__global__ __launch_bounds__(kPlocTail) void k_ploc_tail(PlocArgs a, uint32_t p)
{
    __shared__ float4   s_lo[kPlocTail], s_hi[kPlocTail];
    __shared__ int      s_ref[kPlocTail], s_nn[kPlocTail];
    __shared__ uint32_t s_wave[2][kPlocTail / 64];
    const int      t = (int)threadIdx.x, R = (int)a.radius;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long below = lane ? (~0ull >> (64u - lane)) : 0ull;
    int            m = (int)a.ctrl[2 * p];
    uint32_t       created = a.ctrl[2 * p + 1];
    if (t < m) s_lo[t] = a.lo[p][t], s_hi[t] = a.hi[p][t], s_ref[t] = a.ref[p][t];
    __syncthreads();
    while (m > 1)
    {
        if (t < m) s_nn[t] = ploc_nearest(s_lo, s_hi, 0, t, m, R);
        __syncthreads();
        bool keep, create;
        int  j;
        ploc_flags(s_nn, t, m, keep, create, j);
        const unsigned long long bk = __ballot(keep), bc = __ballot(create);
        if (lane == 0) s_wave[0][wave] = (uint32_t)__popcll(bk), s_wave[1][wave] = (uint32_t)__popcll(bc);
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo, blo = lo, bhi = lo;
        int    ref = 0, bref = 0;
        if (keep) lo = s_lo[t], hi = s_hi[t], ref = s_ref[t];
        if (create) blo = s_lo[j], bhi = s_hi[j], bref = s_ref[j];
        __syncthreads();  // every read of the old array is done
        uint32_t pk = (uint32_t)__popcll(bk & below), pc = (uint32_t)__popcll(bc & below), tk = 0, tc = 0;
        for (uint32_t w = 0; w < kPlocTail / 64; ++w)
        {
            if (w < wave) pk += s_wave[0][w], pc += s_wave[1][w];
            tk += s_wave[0][w], tc += s_wave[1][w];
        }
        if (create)
        {
            const uint32_t node = (a.n - 2u) - (created + pc);
            float4         mlo, mhi;
            ploc_emit(a, node, lo, hi, ref, blo, bhi, bref, mlo, mhi);
            lo = mlo, hi = mhi, ref = (int)node;
        }
        if (keep) s_lo[pk] = lo, s_hi[pk] = hi, s_ref[pk] = ref;
        m = (int)tk, created += tc;
        __syncthreads();
    }
    if (t == 0) *a.max_depth = f2u(s_hi[0].w);
}

__device__ __forceinline__ uint32_t ploc_count_of(const PlocArgs& a, int ref) { return ref < 0 ? 1u : a.count[ref]; }

// depth-first position of every triangle: the triangles to the left of it at each of its ancestors
__global__ __launch_bounds__(kPlocBlock) void k_ploc_leaves(PlocArgs a)
{
    const uint32_t i = blockIdx.x * kPlocBlock + threadIdx.x;
    if (i >= a.n) return;
    uint32_t pos = 0, cur = a.parent[(size_t)(a.n - 1u) + i];
    while (cur != 0xffffffffu)
    {
        const uint32_t node = cur >> 1;
        if (cur & 1u) pos += ploc_count_of(a, a.child[2 * (size_t)node]);
        cur = a.parent[node];
    }
    const uint32_t g = a.order[i];
    a.leaf_pos[i]    = pos;
    a.leaf_tri[pos]  = g;
    for (int k = 0; k < 4; ++k) a.tris_sorted[4 * (size_t)pos + k] = a.tri_raw[4 * (size_t)g + k];
}

// q3 of bvh.hip's node: (child0, child1, traversal child0, traversal child1)
__global__ __launch_bounds__(kPlocBlock) void k_ploc_links(PlocArgs a)
{
    const uint32_t node = blockIdx.x * kPlocBlock + threadIdx.x;
    if (node + 1u >= a.n) return;
    uint32_t link[4];
    for (int s = 0; s < 2; ++s)
    {
        const int      ref = a.child[2 * (size_t)node + s];
        const uint32_t cnt = ploc_count_of(a, ref);
        link[s]            = ref < 0 ? ~a.leaf_pos[~ref] : (uint32_t)ref;
        link[2 + s]        = (uint32_t)ref;
        if (cnt <= (uint32_t)kLeafMax)
        {
            int r = ref;
            while (r >= 0) r = a.child[2 * (size_t)r];  // the leftmost triangle below
            link[2 + s] = ~(a.leaf_pos[~r] | ((cnt - 1u) << kLeafCountShift));
        }
    }
    a.nodes[4 * (size_t)node + 3] = make_float4(u2f(link[0]), u2f(link[1]), u2f(link[2]), u2f(link[3]));
}
}  // namespace

int launch_bvh_build_ploc(hipStream_t stream, const BvhBuildArgs& b, const PlocScratch& s, uint32_t radius)
{
    const uint32_t n = b.tri_count;
    if (n == 0) return 0;
    const int src = launch_bvh_sort(stream, b);  // setup, Morton codes, radix sort: order = b.vals[src]
    PlocArgs  a{};
    a.n = n, a.radius = radius < 1u ? 1u : (radius > kPlocMaxRadius ? kPlocMaxRadius : radius);
    a.tri_box = b.tri_box, a.order = b.vals[src], a.tri_raw = b.tri_raw;
    a.lo[0] = s.boxes, a.lo[1] = s.boxes + n, a.hi[0] = s.boxes + 2 * (size_t)n, a.hi[1] = s.boxes + 3 * (size_t)n;
    a.ref[0] = reinterpret_cast<int*>(b.vals[src ^ 1]), a.ref[1] = reinterpret_cast<int*>(b.keys[src]);  // both free after the sort
    a.nn = reinterpret_cast<int*>(s.ints), a.child = reinterpret_cast<int*>(s.ints + n), a.ctrl = s.ints + 3 * (size_t)n;
    a.blk = b.hist, a.nodes = b.nodes, a.parent = b.parent, a.count = b.keys[src ^ 1], a.leaf_pos = b.flags;
    a.leaf_tri = b.leaf_tri, a.tris_sorted = b.tris_sorted, a.max_depth = b.max_depth;
    const uint32_t blocks = (n + kPlocBlock - 1) / kPlocBlock;
    hipLaunchKernelGGL(k_ploc_init, dim3(blocks), dim3(kPlocBlock), 0, stream, a);
    const uint32_t none = 0xffffffffu;
    if (hipMemcpyAsync(a.parent, &none, sizeof(none), hipMemcpyHostToDevice, stream) != hipSuccess) return 1;  // the root (node 0; the only leaf when n == 1)
    uint32_t m = n, p = 0;
    while (m > kPlocTail)
    {
        const uint32_t mb = (m + kPlocBlock - 1) / kPlocBlock;
        hipLaunchKernelGGL(k_ploc_nn, dim3(mb), dim3(kPlocBlock), 0, stream, a, p);
        hipLaunchKernelGGL(k_ploc_count, dim3(mb), dim3(kPlocBlock), 0, stream, a, p);
        hipLaunchKernelGGL(k_ploc_scan, dim3(1), dim3(1024), 0, stream, a, p);
        hipLaunchKernelGGL(k_ploc_merge, dim3(mb), dim3(kPlocBlock), 0, stream, a, p);
        uint32_t next = 0;
        if (hipMemcpyAsync(&next, a.ctrl + 2 * (p ^ 1u), sizeof(next), hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
        if (hipStreamSynchronize(stream) != hipSuccess) return 1;
        if (next >= m || next == 0) return 2;  // cannot happen: the closest pair of an iteration always merges
        m = next, p ^= 1u;
    }
    if (n >= 2) hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(kPlocTail), 0, stream, a, p);
    hipLaunchKernelGGL(k_ploc_leaves, dim3(blocks), dim3(kPlocBlock), 0, stream, a);
    if (n >= 2) hipLaunchKernelGGL(k_ploc_links, dim3(blocks), dim3(kPlocBlock), 0, stream, a);
    return 0;
}
}  // namespace cap
