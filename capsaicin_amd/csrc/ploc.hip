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
#include "cap_trace.h"

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
    uint32_t*       tag[2];         // sah_device build only (else null): clusters merge only with clusters of the same tag
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

// the same among the clusters that carry cluster i's tag (the sah_device build: a tag is a contiguous run of positions); -1 if
// i is the last cluster of its tag
__device__ __forceinline__ int ploc_nearest_tagged(const float4* t_lo, const float4* t_hi, const uint32_t* t_tag, int tile_first, int i, int m, int radius)
{
    const float4   lo = t_lo[i - tile_first], hi = t_hi[i - tile_first];
    const uint32_t tg = t_tag[i - tile_first];
    float          best = INFINITY;
    int            bj = -1, any = -1;
    const int      j0 = i - radius < 0 ? 0 : i - radius, j1 = i + radius > m - 1 ? m - 1 : i + radius;
    for (int j = j0; j <= j1; ++j)
    {
        if (j == i || t_tag[j - tile_first] != tg) continue;
        if (any < 0 || (j > i ? j - i : i - j) < (any > i ? any - i : i - any)) any = j;
        const float ar = union_half_area(lo, hi, t_lo[j - tile_first], t_hi[j - tile_first]);
        if (ar < best || (ar == best && ploc_tie_key(i, j) < ploc_tie_key(i, bj))) best = ar, bj = j;
    }
    return bj < 0 ? any : bj;  // (every area inf / nan: the closest position of the tag)
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

__global__ __launch_bounds__(kPlocBlock) void k_ploc_nn_tagged(PlocArgs a, uint32_t p)
{
    __shared__ float4   s_lo[kPlocBlock + 2 * kPlocMaxRadius], s_hi[kPlocBlock + 2 * kPlocMaxRadius];
    __shared__ uint32_t s_tag[kPlocBlock + 2 * kPlocMaxRadius];
    const int m = (int)a.ctrl[2 * p], base = (int)(blockIdx.x * kPlocBlock), R = (int)a.radius;
    if (base >= m) return;
    for (int t = (int)threadIdx.x; t < (int)kPlocBlock + 2 * R; t += (int)kPlocBlock)
    {
        const int j = base - R + t;
        if (j >= 0 && j < m) s_lo[t] = a.lo[p][j], s_hi[t] = a.hi[p][j], s_tag[t] = a.tag[p][j];
    }
    __syncthreads();
    const int i = base + (int)threadIdx.x;
    if (i < m) a.nn[i] = ploc_nearest_tagged(s_lo, s_hi, s_tag, base - R, i, m, R);
}

// (keeps its position or becomes a node, becomes a node) of cluster i
__device__ __forceinline__ void ploc_flags(const int* nn, int i, int m, bool& keep, bool& create, int& j)
{
    keep = create = false, j = 0;
    if (i >= m) return;
    j                 = nn[i];
    if (j < 0)  // (tagged search only) the last cluster of its tag: stays
    {
        keep = true, j = i;
        return;
    }
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
    if (a.tag[0]) a.tag[q][pk] = a.tag[p][i];
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

// ================================================================================================================================
// sah_device: binned surface-area splits from the root down, clustering below (CAP_BVH_BUILD_SAH_DEVICE).
//
// The reference asks its driver for PREFER_FAST_TRACE (blas_system.cpp:44, tlas_system.cpp:24).  The clustering build above is the
// better tree below ~ 8 triangles and loses to top-down surface-area splits above that (docs/experiments.md (71): the last merges
// are made inside a +- 16-position window).  So: the cluster array (triangles in Morton order) is split top-down, level by level,
// with the host builder's criterion (sah_builder.cpp: binned SAH over all three axes, cost = area x count per side) until a
// segment holds <= `leaf` clusters; each such segment then gets its subtree from the clustering iterations above, restricted to
// the segment by a tag.  Everything is deterministic: bins are integer atomic max / add, positions come from prefix sums, node
// numbers from the order of the segment list -- every rank of a multi-GPU job builds the same tree.
//
// A level = the list of ACTIVE segments (contiguous position ranges with more than `leaf` clusters; each is one node of the tree):
//   bin      every cluster of an active segment into K bins per axis of the segment's centroid bounds (count + box per bin;
//            a workgroup or wave whose clusters share one segment accumulates in LDS first)
//   split    one thread per segment: sweep the bins, take the cheapest plane (or the median position when there is none, and
//            from depth 36 on: that bounds the depth), write the node's two child boxes
//   scan     over the segments: index and bin storage of the children that stay active
//   emit     the children's segment records / tags
//   flags, scan, scatter: stable partition of every active segment (one prefix sum over the whole array), the children's
//            centroid bounds accumulated on the way
// The stable partition keeps the Morton order inside every segment, which the clustering's position window relies on.
constexpr uint32_t kSahFinal     = 0x80000000u;  // tag bit: the cluster's segment is finished; low bits = (node << 1) | slot
constexpr uint32_t kSahChunk     = 1024;         // clusters per workgroup in the binning pass
constexpr uint32_t kSahMaxBins   = 32;
constexpr uint32_t kSahBinWords  = 7;            // count, max of enc(-lo.xyz), max of enc(hi.xyz)
constexpr uint32_t kSahMedianDepth = 36;         // as sah_builder.cpp

struct SahSeg
{
    uint32_t start, count, node, depth;
    uint32_t cb[6];     // centroid bounds: enc(-lo.xyz), enc(hi.xyz), atomicMax targets (0 = empty)
    uint32_t bin_off;   // first word of the segment's bins
    uint32_t decision;  // axis | bin << 2 | median << 8
    uint32_t n_left;
    uint32_t tag_l, tag_r;
    uint32_t pad;
};
static_assert(sizeof(SahSeg) == 64, "SahSeg");

struct SahArgs
{
    PlocArgs  pl;
    uint32_t  leaf;        // segments of at most this many clusters go to the clustering
    SahSeg*   segs[2];
    uint32_t* bins;
    uint32_t* seg_scan;    // 2 per segment: active children, their bin words
    uint32_t* scan_l;      // per position: (clusters going left before it) << 1 | goes left
    uint32_t* node_depth;  // per top node: nodes on the path from the root
    uint32_t* sctrl;       // [0] active segments of the next level, [1] their bin words
};

// order-preserving float <-> uint (as bvh.hip's); 0 is below every encoding
__device__ __forceinline__ uint32_t sah_enc(float f)
{
    const uint32_t u = f2u(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sah_dec(uint32_t o) { return u2f((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__device__ __forceinline__ uint32_t sah_bins_for(uint32_t count) { return count >= 1024u ? 32u : (count >= 128u ? 16u : 8u); }

// the bin of a centroid coordinate; the same expression wherever a cluster's side is decided
__device__ __forceinline__ uint32_t sah_bin_of(float c, float lo, float hi, uint32_t K)
{
    const float ext = hi - lo;
    if (!(ext > 0.0f)) return 0u;
    const int k = (int)((c - lo) * ((float)K / ext));
    return (uint32_t)(k < 0 ? 0 : (k >= (int)K ? (int)K - 1 : k));
}

struct SahBounds
{
    float lo[3], hi[3];
};
__device__ __forceinline__ SahBounds sah_seg_bounds(const SahSeg& g)
{
    SahBounds b;
    for (int k = 0; k < 3; ++k) b.lo[k] = -sah_dec(g.cb[k]), b.hi[k] = sah_dec(g.cb[3 + k]);
    return b;
}

// 6 values: max over the wave (every lane gets the result)
__device__ __forceinline__ void sah_wave_max6(uint32_t v[6])
{
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 6; ++k)
        {
            const uint32_t o = (uint32_t)__shfl_xor((int)v[k], off);
            v[k]             = v[k] > o ? v[k] : o;
        }
}

// max of 6 values over the workgroup (256 threads); the result is valid in thread 0..5 as v[0] of that thread ... returned through s_out[6]
__device__ __forceinline__ void sah_block_max6(uint32_t v[6], uint32_t (*s_red)[6], uint32_t* s_out)
{
    sah_wave_max6(v);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 6; ++k) s_red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 6u)
    {
        uint32_t m = s_red[0][threadIdx.x];
        for (uint32_t w = 1; w < kPlocBlock / 64u; ++w) m = m > s_red[w][threadIdx.x] ? m : s_red[w][threadIdx.x];
        s_out[threadIdx.x] = m;
    }
    __syncthreads();
}

__global__ __launch_bounds__(kPlocBlock) void k_sah_init(SahArgs x)
{
    __shared__ uint32_t s_red[kPlocBlock / 64][6], s_out[6];
    const PlocArgs& a = x.pl;
    uint32_t        cb[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * kSahChunk + threadIdx.x; i < min(a.n, (blockIdx.x + 1u) * kSahChunk); i += kPlocBlock)
    {
        const uint32_t g  = a.order[i];
        float4         lo = a.tri_box[2 * (size_t)g], hi = a.tri_box[2 * (size_t)g + 1];
        float*         l  = &lo.x;
        float*         h  = &hi.x;
        for (int k = 0; k < 3; ++k)
        {
            const float pad = 1e-5f * fmaxf(1.0f, fmaxf(fabsf(l[k]), fabsf(h[k])));  // the leaf pad of bvh.hip's refit, as k_ploc_init
            l[k] -= pad, h[k] += pad;
            const float    c  = 0.5f * (l[k] + h[k]);
            const uint32_t en = sah_enc(-c), ep = sah_enc(c);
            cb[k] = cb[k] > en ? cb[k] : en, cb[3 + k] = cb[3 + k] > ep ? cb[3 + k] : ep;
        }
        lo.w = u2f(1u), hi.w = u2f(0u);
        a.lo[0][i] = lo, a.hi[0][i] = hi, a.ref[0][i] = ~(int)i, a.tag[0][i] = 0u;
    }
    sah_block_max6(cb, s_red, s_out);
    if (threadIdx.x < 6u) atomicMax(&x.segs[0][0].cb[threadIdx.x], s_out[threadIdx.x]);
}

// one cluster into a set of bins (LDS or global)
template <bool GLOBAL>
__device__ __forceinline__ void sah_bin_add(uint32_t* bins, const SahBounds& sb, uint32_t K, const float4& lo, const float4& hi)
{
    const float    c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
    const uint32_t e[6] = {sah_enc(-lo.x), sah_enc(-lo.y), sah_enc(-lo.z), sah_enc(hi.x), sah_enc(hi.y), sah_enc(hi.z)};
    for (int ax = 0; ax < 3; ++ax)
    {
        uint32_t* q = bins + ((uint32_t)ax * K + sah_bin_of(c[ax], sb.lo[ax], sb.hi[ax], K)) * kSahBinWords;
        atomicAdd(q, 1u);
        for (int k = 0; k < 6; ++k) atomicMax(q + 1 + k, e[k]);
    }
}

__global__ __launch_bounds__(kPlocBlock) void k_sah_bin(SahArgs x, uint32_t p)
{
    __shared__ uint32_t s_bins[4 * 3 * kSahMaxBins * kSahBinWords];
    const PlocArgs& a  = x.pl;
    const uint32_t  c0 = blockIdx.x * kSahChunk, c1 = min(a.n, c0 + kSahChunk);
    const uint32_t  t_first = a.tag[p][c0], t_last = a.tag[p][c1 - 1u];
    const uint32_t  lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (t_first == t_last)
    {
        // segments are contiguous: the whole chunk belongs to one
        if (t_first & kSahFinal) return;
        const SahSeg    g  = x.segs[p][t_first];
        const SahBounds sb = sah_seg_bounds(g);
        const uint32_t  K = sah_bins_for(g.count), words = 3u * K * kSahBinWords;
        for (uint32_t w = threadIdx.x; w < words; w += kPlocBlock) s_bins[w] = 0u;
        __syncthreads();
        for (uint32_t i = c0 + threadIdx.x; i < c1; i += kPlocBlock) sah_bin_add<false>(s_bins, sb, K, a.lo[p][i], a.hi[p][i]);
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < words; w += kPlocBlock)
        {
            const uint32_t v = s_bins[w];
            if (v == 0u) continue;
            if (w % kSahBinWords == 0u)
                atomicAdd(x.bins + g.bin_off + w, v);
            else
                atomicMax(x.bins + g.bin_off + w, v);
        }
        return;
    }
    // a mixed chunk: wave by wave, run by run (a run = the wave's clusters of one segment; segments are contiguous, so a wave holds few).
    // A run of a dozen clusters or more accumulates in the wave's own bins in LDS and flushes one atomic per touched word; shorter
    // runs use the atomics directly (21 per cluster).
    uint32_t* const wb = s_bins + wave * (3u * kSahMaxBins * kSahBinWords);
    for (uint32_t r0 = c0 + wave * 64u; r0 < c1; r0 += kPlocBlock)
    {
        const uint32_t     i = r0 + lane;
        const uint32_t     t = i < c1 ? a.tag[p][i] : kSahFinal;
        unsigned long long todo = __ballot(!(t & kSahFinal));
        while (todo != 0ull)
        {
            const int                leader = __ffsll((long long)todo) - 1;
            const uint32_t           cur    = (uint32_t)__shfl((int)t, leader);
            const unsigned long long same   = __ballot(t == cur) & todo;
            todo &= ~same;
            const bool      mine = ((same >> lane) & 1ull) != 0ull;
            const SahSeg    g    = x.segs[p][cur];
            const SahBounds sb   = sah_seg_bounds(g);
            const uint32_t  K    = sah_bins_for(g.count);
            if (__popcll(same) >= 12)
            {
                const uint32_t words = 3u * K * kSahBinWords;
                for (uint32_t w = lane; w < words; w += 64u) wb[w] = 0u;
                wave_handoff();
                if (mine) sah_bin_add<false>(wb, sb, K, a.lo[p][i], a.hi[p][i]);
                wave_handoff();
                for (uint32_t w = lane; w < words; w += 64u)
                {
                    const uint32_t v = wb[w];
                    if (v == 0u) continue;
                    if (w % kSahBinWords == 0u)
                        atomicAdd(x.bins + g.bin_off + w, v);
                    else
                        atomicMax(x.bins + g.bin_off + w, v);
                }
                wave_handoff();
            }
            else if (mine)
                sah_bin_add<true>(x.bins + g.bin_off, sb, K, a.lo[p][i], a.hi[p][i]);
        }
    }
}

struct SahBox
{
    float lo[3], hi[3];
    __device__ void reset()
    {
        for (int k = 0; k < 3; ++k) lo[k] = INFINITY, hi[k] = -INFINITY;
    }
    __device__ void grow_bin(const uint32_t* q)  // q: a non-empty bin's words 1..6
    {
        for (int k = 0; k < 3; ++k) lo[k] = fminf(lo[k], -sah_dec(q[k])), hi[k] = fmaxf(hi[k], sah_dec(q[3 + k]));
    }
    __device__ float half_area() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

__global__ __launch_bounds__(64) void k_sah_split(SahArgs x, uint32_t p, uint32_t n_seg)
{
    const PlocArgs& a = x.pl;
    const uint32_t  s = blockIdx.x * 64u + threadIdx.x;
    if (s >= n_seg) return;
    SahSeg          g  = x.segs[p][s];
    const SahBounds sb = sah_seg_bounds(g);
    const uint32_t  K  = sah_bins_for(g.count);
    const uint32_t* B  = x.bins + g.bin_off;
    int             best_axis = -1;
    uint32_t        best_bin = 0, best_left = 0;
    float           best_cost = INFINITY;
    if (g.depth < kSahMedianDepth)
        for (int ax = 0; ax < 3; ++ax)
        {
            if (!(sb.hi[ax] - sb.lo[ax] > 0.0f)) continue;
            const uint32_t* A = B + (uint32_t)ax * K * kSahBinWords;
            float           right_area[kSahMaxBins];
            uint32_t        right_cnt[kSahMaxBins];
            SahBox          acc;
            acc.reset();
            uint32_t c = 0;
            for (int k = (int)K - 1; k > 0; --k)
            {
                const uint32_t* q = A + (uint32_t)k * kSahBinWords;
                if (q[0]) acc.grow_bin(q + 1), c += q[0];
                right_area[k] = c ? acc.half_area() : 0.0f, right_cnt[k] = c;
            }
            acc.reset();
            c = 0;
            for (int k = 0; k + 1 < (int)K; ++k)
            {
                const uint32_t* q = A + (uint32_t)k * kSahBinWords;
                if (q[0]) acc.grow_bin(q + 1), c += q[0];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                const float cost = acc.half_area() * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
                if (cost < best_cost) best_cost = cost, best_axis = ax, best_bin = (uint32_t)k, best_left = c;
            }
        }
    SahBox box[2];
    box[0].reset(), box[1].reset();
    if (best_axis >= 0)
    {
        const uint32_t* A = B + (uint32_t)best_axis * K * kSahBinWords;
        for (uint32_t k = 0; k < K; ++k)
            if (A[k * kSahBinWords]) box[k <= best_bin ? 0 : 1].grow_bin(A + k * kSahBinWords + 1);
        g.decision = (uint32_t)best_axis | (best_bin << 2);
        g.n_left   = best_left;
    }
    else
    {
        // no plane separates the centroids (or the depth limit): halves by position; both children get the segment's box
        for (uint32_t k = 0; k < K; ++k)
            if (B[k * kSahBinWords]) box[0].grow_bin(B + k * kSahBinWords + 1);
        box[1]     = box[0];
        g.decision = 1u << 8;
        g.n_left   = g.count / 2u;
    }
    float4* q = a.nodes + 4 * (size_t)g.node;
    q[0]      = make_float4(box[0].lo[0], box[0].lo[1], box[0].lo[2], box[0].hi[0]);
    q[1]      = make_float4(box[0].hi[1], box[0].hi[2], box[1].lo[0], box[1].lo[1]);
    q[2]      = make_float4(box[1].lo[2], box[1].hi[0], box[1].hi[1], box[1].hi[2]);
    a.count[g.node]      = g.count;
    x.node_depth[g.node] = g.depth;
    x.segs[p][s].decision = g.decision, x.segs[p][s].n_left = g.n_left;
    const uint32_t nl = g.n_left, nr = g.count - g.n_left;
    x.seg_scan[2 * s]     = (nl > x.leaf ? 1u : 0u) + (nr > x.leaf ? 1u : 0u);
    x.seg_scan[2 * s + 1] = (nl > x.leaf ? 3u * sah_bins_for(nl) * kSahBinWords : 0u) + (nr > x.leaf ? 3u * sah_bins_for(nr) * kSahBinWords : 0u);
}

// exclusive scan of n pairs, one workgroup; totals to out[0], out[1]
__global__ __launch_bounds__(1024) void k_sah_scan_pairs(uint32_t* v, uint32_t n, uint32_t* out)
{
    __shared__ uint32_t s_part[2][1024];
    const uint32_t t = threadIdx.x, per = (n + 1023u) / 1024u, b0 = min(n, t * per), b1 = min(n, b0 + per);
    uint32_t       s0 = 0, s1 = 0;
    for (uint32_t b = b0; b < b1; ++b) s0 += v[2 * b], s1 += v[2 * b + 1];
    s_part[0][t] = s0, s_part[1][t] = s1;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1)
    {
        const uint32_t v0 = t >= off ? s_part[0][t - off] : 0u, v1 = t >= off ? s_part[1][t - off] : 0u;
        __syncthreads();
        s_part[0][t] += v0, s_part[1][t] += v1;
        __syncthreads();
    }
    uint32_t e0 = s_part[0][t] - s0, e1 = s_part[1][t] - s1;
    for (uint32_t b = b0; b < b1; ++b)
    {
        const uint32_t k0 = v[2 * b], k1 = v[2 * b + 1];
        v[2 * b] = e0, v[2 * b + 1] = e1;
        e0 += k0, e1 += k1;
    }
    if (t == 1023u) out[0] = s_part[0][t], out[1] = s_part[1][t];
}

__global__ __launch_bounds__(64) void k_sah_emit(SahArgs x, uint32_t p, uint32_t n_seg, uint32_t node_base_next)
{
    const PlocArgs& a = x.pl;
    const uint32_t  s = blockIdx.x * 64u + threadIdx.x;
    if (s >= n_seg) return;
    const SahSeg g   = x.segs[p][s];
    uint32_t     idx = x.seg_scan[2 * s], off = x.seg_scan[2 * s + 1];
    uint32_t     tags[2];
    for (uint32_t side = 0; side < 2; ++side)
    {
        const uint32_t cnt = side ? g.count - g.n_left : g.n_left, start = side ? g.start + g.n_left : g.start;
        if (cnt > x.leaf)
        {
            SahSeg c{};
            c.start = start, c.count = cnt, c.node = node_base_next + idx, c.depth = g.depth + 1u, c.bin_off = off;
            x.segs[p ^ 1u][idx]          = c;
            a.child[2 * (size_t)g.node + side] = (int)c.node;
            a.parent[c.node]             = (g.node << 1) | side;
            tags[side]                   = idx;
            idx += 1u, off += 3u * sah_bins_for(cnt) * kSahBinWords;
        }
        else
            tags[side] = kSahFinal | (g.node << 1) | side;
    }
    x.segs[p][s].tag_l = tags[0], x.segs[p][s].tag_r = tags[1];
}

// does cluster i of an active segment go to the left child?
__device__ __forceinline__ bool sah_goes_left(const SahSeg& g, uint32_t i, const float4& lo, const float4& hi)
{
    if (g.decision >> 8) return i - g.start < g.n_left;
    const uint32_t  ax = g.decision & 3u, bin = (g.decision >> 2) & 63u;
    const SahBounds sb = sah_seg_bounds(g);
    const float     c  = ax == 0 ? 0.5f * (lo.x + hi.x) : (ax == 1 ? 0.5f * (lo.y + hi.y) : 0.5f * (lo.z + hi.z));
    return sah_bin_of(c, sb.lo[ax], sb.hi[ax], sah_bins_for(g.count)) <= bin;
}

__global__ __launch_bounds__(kPlocBlock) void k_sah_flags(SahArgs x, uint32_t p)
{
    __shared__ uint32_t s_sum;
    const PlocArgs& a = x.pl;
    const uint32_t  i = blockIdx.x * kPlocBlock + threadIdx.x;
    if (threadIdx.x == 0) s_sum = 0u;
    __syncthreads();
    bool left = false;
    if (i < a.n)
    {
        const uint32_t t = a.tag[p][i];
        if (!(t & kSahFinal)) left = sah_goes_left(x.segs[p][t], i, a.lo[p][i], a.hi[p][i]);
        x.scan_l[i] = left ? 1u : 0u;
    }
    const uint32_t nl = (uint32_t)__popcll(__ballot(left));
    if ((threadIdx.x & 63u) == 0 && nl) atomicAdd(&s_sum, nl);
    __syncthreads();
    if (threadIdx.x == 0) a.blk[2 * blockIdx.x] = s_sum, a.blk[2 * blockIdx.x + 1] = 0u;
}

__global__ __launch_bounds__(kPlocBlock) void k_sah_scan_values(SahArgs x)
{
    __shared__ uint32_t s_wave[kPlocBlock / 64];
    const PlocArgs& a = x.pl;
    const uint32_t  i = blockIdx.x * kPlocBlock + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool      left = i < a.n && x.scan_l[i] != 0u;
    const unsigned long long b = __ballot(left), below = lane ? (~0ull >> (64u - lane)) : 0ull;
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t pre = a.blk[2 * blockIdx.x] + (uint32_t)__popcll(b & below);
    for (uint32_t w = 0; w < wave; ++w) pre += s_wave[w];
    if (i < a.n) x.scan_l[i] = (pre << 1) | (left ? 1u : 0u);
}

// Stable partition of every active segment + the centroid bounds of the children that stay active.  kSahChunk positions per workgroup.
// The bounds are 6 atomic max per child: at the top of the tree millions of clusters feed the same two children, so a workgroup whose
// chunk lies in one segment reduces both children's bounds in LDS first (12 atomics per 1024 clusters), a wave whose 64 clusters lie in
// one segment by shuffles (12 per 64); only clusters of mixed waves use the atomics directly.  (The first version reduced only waves
// whose clusters all went to ONE child -- never true at the top, where left and right alternate: 3.7 ms per level at 16.8 M triangles.)
__global__ __launch_bounds__(kPlocBlock) void k_sah_scatter(SahArgs x, uint32_t p)
{
    __shared__ uint32_t s_red[kPlocBlock / 64][12];
    const PlocArgs& a  = x.pl;
    const uint32_t  q  = p ^ 1u, c0 = blockIdx.x * kSahChunk, c1 = min(a.n, c0 + kSahChunk);
    const uint32_t  lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t  t_first = a.tag[p][c0], t_last = a.tag[p][c1 - 1u];
    const bool      uniform = t_first == t_last;  // segments are contiguous: the whole chunk is one segment's
    if (uniform && (t_first & kSahFinal))
    {
        for (uint32_t i = c0 + threadIdx.x; i < c1; i += kPlocBlock)
            a.lo[q][i] = a.lo[p][i], a.hi[q][i] = a.hi[p][i], a.ref[q][i] = a.ref[p][i], a.tag[q][i] = t_first;
        return;
    }
    uint32_t acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // uniform chunks: left child's bounds, right child's bounds
    for (uint32_t i0 = c0 + wave * 64u; i0 < c1; i0 += kPlocBlock)
    {
        const uint32_t i = i0 + lane;
        const bool     in = i < c1;
        const uint32_t t = in ? a.tag[p][i] : kSahFinal;
        uint32_t       cb[6] = {0, 0, 0, 0, 0, 0};
        uint32_t       child = kSahFinal;
        bool           left  = false;
        if (in)
        {
            const float4 lo = a.lo[p][i], hi = a.hi[p][i];
            const int    ref = a.ref[p][i];
            uint32_t     dest = i, tag = t;
            if (!(t & kSahFinal))
            {
                const SahSeg   g = x.segs[p][t];
                const uint32_t v = x.scan_l[i], r = (v >> 1) - (x.scan_l[g.start] >> 1);
                left = (v & 1u) != 0u;
                if (left)
                    dest = g.start + r, tag = g.tag_l;
                else
                    dest = g.start + g.n_left + ((i - g.start) - r), tag = g.tag_r;
                child = tag;
                if (!(child & kSahFinal))
                {
                    const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
                    for (int k = 0; k < 3; ++k) cb[k] = sah_enc(-c[k]), cb[3 + k] = sah_enc(c[k]);
                }
            }
            a.lo[q][dest] = lo, a.hi[q][dest] = hi, a.ref[q][dest] = ref, a.tag[q][dest] = tag;
        }
        if (uniform)
        {
            for (int k = 0; k < 6; ++k)
            {
                const uint32_t l = left ? cb[k] : 0u, r = left ? 0u : cb[k];
                acc[k] = acc[k] > l ? acc[k] : l, acc[6 + k] = acc[6 + k] > r ? acc[6 + k] : r;
            }
            continue;
        }
        // a mixed chunk: wave by wave
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (__ballot(t != first) == 0ull)
        {
            if (first & kSahFinal) continue;
            const SahSeg g = x.segs[p][first];
            uint32_t     l6[6], r6[6];
            for (int k = 0; k < 6; ++k) l6[k] = left ? cb[k] : 0u, r6[k] = left ? 0u : cb[k];
            sah_wave_max6(l6), sah_wave_max6(r6);
            if (lane < 6u && !(g.tag_l & kSahFinal)) atomicMax(&x.segs[q][g.tag_l].cb[lane], l6[lane]);
            if (lane >= 6u && lane < 12u && !(g.tag_r & kSahFinal)) atomicMax(&x.segs[q][g.tag_r].cb[lane - 6u], r6[lane - 6u]);
        }
        else if (!(child & kSahFinal))
            for (int k = 0; k < 6; ++k) atomicMax(&x.segs[q][child].cb[k], cb[k]);
    }
    if (!uniform) return;
    sah_wave_max6(acc), sah_wave_max6(acc + 6);
    if (lane == 0)
        for (int k = 0; k < 12; ++k) s_red[wave][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 12u)
    {
        uint32_t m = 0;
        for (uint32_t w = 0; w < kPlocBlock / 64u; ++w) m = m > s_red[w][threadIdx.x] ? m : s_red[w][threadIdx.x];
        const SahSeg   g   = x.segs[p][t_first];
        const uint32_t tag = threadIdx.x < 6u ? g.tag_l : g.tag_r;
        if (!(tag & kSahFinal) && m != 0u) atomicMax(&x.segs[q][tag].cb[threadIdx.x % 6u], m);
    }
}

// after the clustering: the clusters left are the finished segments' subtrees (or single triangles); hang them into the top tree
__global__ __launch_bounds__(kPlocBlock) void k_sah_patch(SahArgs x, uint32_t p, uint32_t m)
{
    const PlocArgs& a = x.pl;
    const uint32_t  i = blockIdx.x * kPlocBlock + threadIdx.x;
    if (i >= m) return;
    const uint32_t t = a.tag[p][i] & ~kSahFinal, node = t >> 1, slot = t & 1u;
    const int      ref = a.ref[p][i];
    a.child[2 * (size_t)node + slot] = ref;
    a.parent[ref >= 0 ? (size_t)ref : (size_t)(a.n - 1u) + (size_t)~ref] = (node << 1) | slot;
    atomicMax(a.max_depth, x.node_depth[node] + f2u(a.hi[p][i].w));
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

namespace cap
{
namespace
{
constexpr uint32_t kSahMinLeaf = 16;  // bounds the segment list (n / leaf entries) and the bin storage (16 words per triangle)
inline size_t up4(size_t w) { return (w + 3) & ~(size_t)3; }
inline size_t sah_max_segs(uint32_t n) { return (size_t)n / kSahMinLeaf + 16; }
}  // namespace

size_t bvh_sah_device_scratch_words(uint32_t n)
{
    const size_t ms = sah_max_segs(n);
    return 2 * up4(n) + up4(n) + up4(n) + up4(2 * ms) + 16 + 2 * ms * (sizeof(SahSeg) / 4) + 16 * (size_t)n + 3 * kSahMaxBins * kSahBinWords + 64;
}

int launch_bvh_build_sah_device(hipStream_t stream, const BvhBuildArgs& b, const PlocScratch& s, uint32_t* scratch, uint32_t radius, uint32_t leaf)
{
    const uint32_t n = b.tri_count;
    if (n == 0) return 0;
    leaf = leaf < kSahMinLeaf ? kSahMinLeaf : leaf;
    if (n <= leaf || n < 2) return launch_bvh_build_ploc(stream, b, s, radius);  // nothing to split: the clustering alone
    const int src = launch_bvh_sort(stream, b);
    SahArgs   x{};
    PlocArgs& a = x.pl;
    a.n = n, a.radius = radius < 1u ? 1u : (radius > kPlocMaxRadius ? kPlocMaxRadius : radius);
    a.tri_box = b.tri_box, a.order = b.vals[src], a.tri_raw = b.tri_raw;
    a.lo[0] = s.boxes, a.lo[1] = s.boxes + n, a.hi[0] = s.boxes + 2 * (size_t)n, a.hi[1] = s.boxes + 3 * (size_t)n;
    a.ref[0] = reinterpret_cast<int*>(b.vals[src ^ 1]), a.ref[1] = reinterpret_cast<int*>(b.keys[src]);
    a.nn = reinterpret_cast<int*>(s.ints), a.child = reinterpret_cast<int*>(s.ints + n), a.ctrl = s.ints + 3 * (size_t)n;
    a.blk = b.hist, a.nodes = b.nodes, a.parent = b.parent, a.count = b.keys[src ^ 1], a.leaf_pos = b.flags;
    a.leaf_tri = b.leaf_tri, a.tris_sorted = b.tris_sorted, a.max_depth = b.max_depth;
    const size_t ms = sah_max_segs(n);
    uint32_t*    w  = scratch;
    a.tag[0] = w, w += up4(n);
    a.tag[1] = w, w += up4(n);
    x.scan_l = w, w += up4(n);
    x.node_depth = w, w += up4(n);
    x.seg_scan = w, w += up4(2 * ms);
    x.sctrl = w, w += 16;
    x.segs[0] = reinterpret_cast<SahSeg*>(w), w += ms * (sizeof(SahSeg) / 4);
    x.segs[1] = reinterpret_cast<SahSeg*>(w), w += ms * (sizeof(SahSeg) / 4);
    x.bins = w;
    x.leaf = leaf;

    const uint32_t blocks = (n + kPlocBlock - 1) / kPlocBlock;
    SahSeg         root{};
    root.count = n, root.depth = 1u;
    if (hipMemcpyAsync(x.segs[0], &root, sizeof(root), hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    const uint32_t none = 0xffffffffu;
    if (hipMemcpyAsync(a.parent, &none, sizeof(none), hipMemcpyHostToDevice, stream) != hipSuccess) return 1;  // the root
    hipLaunchKernelGGL(k_sah_init, dim3((n + kSahChunk - 1) / kSahChunk), dim3(kPlocBlock), 0, stream, x);

    // ---- top-down levels ----
    uint32_t n_seg = 1, node_base = 0, p = 0, bin_words = 3u * (n >= 1024u ? 32u : (n >= 128u ? 16u : 8u)) * kSahBinWords;
    for (uint32_t level = 0; n_seg > 0; ++level)
    {
        if (level > 4096u || n_seg > ms || (size_t)bin_words > 16 * (size_t)n + 3 * kSahMaxBins * kSahBinWords) return 4;
        if (hipMemsetAsync(x.bins, 0, sizeof(uint32_t) * (size_t)bin_words, stream) != hipSuccess) return 1;
        hipLaunchKernelGGL(k_sah_bin, dim3((n + kSahChunk - 1) / kSahChunk), dim3(kPlocBlock), 0, stream, x, p);
        const uint32_t sb = (n_seg + 63u) / 64u;
        hipLaunchKernelGGL(k_sah_split, dim3(sb), dim3(64), 0, stream, x, p, n_seg);
        hipLaunchKernelGGL(k_sah_scan_pairs, dim3(1), dim3(1024), 0, stream, x.seg_scan, n_seg, x.sctrl);
        hipLaunchKernelGGL(k_sah_emit, dim3(sb), dim3(64), 0, stream, x, p, n_seg, node_base + n_seg);
        hipLaunchKernelGGL(k_sah_flags, dim3(blocks), dim3(kPlocBlock), 0, stream, x, p);
        hipLaunchKernelGGL(k_sah_scan_pairs, dim3(1), dim3(1024), 0, stream, a.blk, blocks, x.sctrl + 2);
        hipLaunchKernelGGL(k_sah_scan_values, dim3(blocks), dim3(kPlocBlock), 0, stream, x);
        hipLaunchKernelGGL(k_sah_scatter, dim3((n + kSahChunk - 1) / kSahChunk), dim3(kPlocBlock), 0, stream, x, p);
        uint32_t next[2] = {0, 0};
        if (hipMemcpyAsync(next, x.sctrl, sizeof(next), hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
        if (hipStreamSynchronize(stream) != hipSuccess) return 1;
        node_base += n_seg, n_seg = next[0], bin_words = next[1], p ^= 1u;
    }
    const uint32_t top_nodes = node_base;

    // ---- the clustering inside the finished segments ----
    uint32_t ctrl0[2] = {n, 0u};
    if (hipMemcpyAsync(a.ctrl + 2 * p, ctrl0, sizeof(ctrl0), hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    uint32_t m = n;
    for (uint32_t it = 0;; ++it)
    {
        if (it > 4096u) return 2;
        const uint32_t mb = (m + kPlocBlock - 1) / kPlocBlock;
        hipLaunchKernelGGL(k_ploc_nn_tagged, dim3(mb), dim3(kPlocBlock), 0, stream, a, p);
        hipLaunchKernelGGL(k_ploc_count, dim3(mb), dim3(kPlocBlock), 0, stream, a, p);
        hipLaunchKernelGGL(k_ploc_scan, dim3(1), dim3(1024), 0, stream, a, p);
        hipLaunchKernelGGL(k_ploc_merge, dim3(mb), dim3(kPlocBlock), 0, stream, a, p);
        uint32_t next = 0;
        if (hipMemcpyAsync(&next, a.ctrl + 2 * (p ^ 1u), sizeof(next), hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
        if (hipStreamSynchronize(stream) != hipSuccess) return 1;
        if (next > m || next == 0) return 2;
        const bool done = next == m;  // nothing merged: every segment is down to one cluster
        m = next, p ^= 1u;
        if (done) break;
    }
    if (m != top_nodes + 1u) return 3;  // a binary tree over m finished segments has m - 1 inner nodes
    hipLaunchKernelGGL(k_sah_patch, dim3((m + kPlocBlock - 1) / kPlocBlock), dim3(kPlocBlock), 0, stream, x, p, m);
    hipLaunchKernelGGL(k_ploc_leaves, dim3(blocks), dim3(kPlocBlock), 0, stream, a);
    hipLaunchKernelGGL(k_ploc_links, dim3(blocks), dim3(kPlocBlock), 0, stream, a);
    return 0;
}
}  // namespace cap
