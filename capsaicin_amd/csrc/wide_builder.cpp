// wide_builder.cpp — binary traversal tree -> compressed 8-wide nodes (cap_wide.h).
//
// Collapse: a wide node stands for a binary node R; its children start as R's two children and the inner one with the largest
// surface area is replaced by its own two children until eight slots are used or nothing is left to open (subtrees of at most
// kWideLeafMax triangles are leaf children; with slots to spare they are opened too, largest first: tighter boxes for free).
//
// Error budget of the padding (what makes the quantised fp32 slab test of trace8.hip conservative).  The kernel computes, per
// axis, t~ = fma(q, S * I, fma(p, I, -(o * I))) with I = rcp(d) (1 ulp), S a power of two; the exact plane is P = p + q * S and
// its exact parameter (P - o) / d.  Collecting the roundings, |t~ - (P - o) / d| <= eps * |I| * (|o| + |p - o| + 2 |P - o|) with
// eps = 2^-23.  For ray origins inside the scene bounds every term is at most max(extent, largest |coordinate|) =: M, so the
// error is below 4 eps M |I| ~ 4.8e-7 M |I|.  A child box grown by kWidePad * M = 4e-6 M on every side therefore satisfies
// t~_near <= t0 <= t~_far on every axis for every true hit point o + t0 d inside the original box, which is all the traversal
// needs (the hit rule itself never looks at boxes).  Directions with |d| < 1e-20 are replaced by +-1e-20 in the box test only:
// such a ray moves less than 1e-15 along that axis over the whole parameter range, far inside the padding.
#include "wide_builder.h"

#include "cap_wide.h"
#include "../../include/capsaicin_scene.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace cap
{
namespace
{
struct Box
{
    double lo[3], hi[3];
    double half_area() const
    {
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};
struct Ref
{
    int32_t  node;   // >= 0: binary internal node, < 0: ~(leaf-order triangle)
    uint32_t count;  // triangles below
    Box      box;
};

struct Collapser
{
    const float*          bn;
    uint32_t              n;
    std::vector<uint32_t> count;  // triangles below each binary internal node
    double                pad;
    WideTree&             out;

    int32_t child_of(uint32_t node, int s) const
    {
        int32_t c;
        std::memcpy(&c, bn + 16 * (size_t)node + 12 + s, 4);
        return c;
    }
    Ref ref_of(uint32_t node, int s) const
    {
        Ref          r;
        const float* q = bn + 16 * (size_t)node + 6 * s;
        for (int k = 0; k < 3; ++k) r.box.lo[k] = q[k], r.box.hi[k] = q[3 + k];
        r.node  = child_of(node, s);
        r.count = r.node >= 0 ? count[(size_t)r.node] : 1u;
        return r;
    }
    void count_subtrees()
    {
        count.assign(n > 1 ? n - 1 : 0, 0u);
        if (n < 2) return;
        // post-order without recursion (the device LBVH numbers its nodes in no particular order)
        std::vector<std::pair<uint32_t, int>> st;
        st.push_back({0u, 0});
        while (!st.empty())
        {
            auto& top = st.back();
            if (top.second < 2)
            {
                const int32_t c = child_of(top.first, top.second++);
                if (c >= 0) st.push_back({(uint32_t)c, 0});
            }
            else
            {
                uint32_t sum = 0;
                for (int s = 0; s < 2; ++s)
                {
                    const int32_t c = child_of(top.first, s);
                    sum += c >= 0 ? count[(size_t)c] : 1u;
                }
                count[top.first] = sum;
                st.pop_back();
            }
        }
    }
    void leaf_triangles(const Ref& r, std::vector<uint32_t>& tris) const
    {
        if (r.node < 0)
        {
            tris.push_back((uint32_t)~r.node);
            return;
        }
        for (int s = 0; s < 2; ++s) leaf_triangles(ref_of((uint32_t)r.node, s), tris);
    }

    // children of the wide node that stands for binary node `root` (or for the whole one-triangle scene)
    void open(const std::vector<Ref>& start, std::vector<Ref>& kids) const
    {
        kids = start;
        for (int pass = 0; pass < 2; ++pass)
        {
            // pass 0 opens inner children (more than kWideLeafMax triangles), pass 1 multi-triangle leaf children
            while (kids.size() < 8)
            {
                int    best = -1;
                double best_area = -1.0;
                for (size_t i = 0; i < kids.size(); ++i)
                {
                    const bool inner = kids[i].count > kWideLeafMax;
                    const bool can   = pass == 0 ? inner : (!inner && kids[i].count > 1);
                    if (can && kids[i].box.half_area() > best_area) best_area = kids[i].box.half_area(), best = (int)i;
                }
                if (best < 0) break;
                const Ref r = kids[(size_t)best];
                kids[(size_t)best] = ref_of((uint32_t)r.node, 0);
                kids.push_back(ref_of((uint32_t)r.node, 1));
            }
        }
    }

    void emit(uint32_t index, const std::vector<Ref>& kids_in, std::vector<std::pair<uint32_t, Ref>>& next_level)
    {
        std::vector<Ref> kids = kids_in;
        const size_t     nk   = kids.size();
        // padded child boxes and the node box
        Box nb;
        for (int k = 0; k < 3; ++k) nb.lo[k] = INFINITY, nb.hi[k] = -INFINITY;
        for (auto& c : kids)
            for (int k = 0; k < 3; ++k)
            {
                c.box.lo[k] -= pad, c.box.hi[k] += pad;
                nb.lo[k] = std::min(nb.lo[k], c.box.lo[k]), nb.hi[k] = std::max(nb.hi[k], c.box.hi[k]);
            }
        // slot assignment: greedy maximum of <child centre - node centre, direction(slot)>
        int  slot_of[8];
        bool slot_used[8] = {false}, kid_done[8] = {false};
        for (size_t step = 0; step < nk; ++step)
        {
            double best = -INFINITY;
            int    bi = -1, bs = -1;
            for (size_t i = 0; i < nk; ++i)
            {
                if (kid_done[i]) continue;
                for (int s = 0; s < 8; ++s)
                {
                    if (slot_used[s]) continue;
                    double v = 0.0;
                    for (int k = 0; k < 3; ++k)
                    {
                        const double off = 0.5 * (kids[i].box.lo[k] + kids[i].box.hi[k]) - 0.5 * (nb.lo[k] + nb.hi[k]);
                        v += ((s >> k) & 1) ? off : -off;
                    }
                    if (v > best) best = v, bi = (int)i, bs = s;
                }
            }
            slot_of[bi] = bs, slot_used[bs] = true, kid_done[bi] = true;
        }
        int kid_at[8];
        for (int s = 0; s < 8; ++s) kid_at[s] = -1;
        for (size_t i = 0; i < nk; ++i) kid_at[slot_of[i]] = (int)i;

        uint32_t* w = out.nodes.data() + (size_t)index * kWideNodeWords;
        std::memset(w, 0, sizeof(uint32_t) * kWideNodeWords);
        // grid origin: the node's low corner rounded down to float
        float p[3];
        for (int k = 0; k < 3; ++k)
        {
            p[k] = (float)nb.lo[k];
            if ((double)p[k] > nb.lo[k]) p[k] = std::nextafterf(p[k], -INFINITY);
            std::memcpy(&w[k], &p[k], 4);
        }
        // grid step per axis: the smallest power of two with ceil((hi - p) / step) <= 255
        double   step[3];
        uint32_t eb[3];
        for (int k = 0; k < 3; ++k)
        {
            const double ext = nb.hi[k] - (double)p[k];
            int          e   = -100;
            if (ext > 0.0)
            {
                int fe;
                (void)std::frexp(ext / 255.0, &fe);  // ext / 255 = m * 2^fe, m in [0.5, 1)
                e = std::max(fe - 1, -100);
            }
            while (std::ceil(ext / std::ldexp(1.0, e)) > 255.0) ++e;
            step[k] = std::ldexp(1.0, e);
            eb[k]   = (uint32_t)(e + 127);
        }
        w[3] = eb[0] << 23;  // the three steps as floats; y and z as their upper halves in w7
        w[7] = ((eb[1] << 23) & 0xffff0000u) | ((eb[2] << 23) >> 16);
        // children: inner ones get consecutive node indices in slot order, leaf ones their triangles in (k, slot) order
        uint32_t imask = 0, tvalid = 0;
        std::vector<uint32_t> leaf_tris[8];
        for (int s = 0; s < 8; ++s)
        {
            if (kid_at[s] < 0) continue;
            const Ref& c = kids[(size_t)kid_at[s]];
            if (c.count > kWideLeafMax)
                imask |= 1u << s;
            else
            {
                leaf_triangles(kids_in[(size_t)kid_at[s]], leaf_tris[s]);
                for (uint32_t k = 0; k < (uint32_t)leaf_tris[s].size(); ++k) tvalid |= 1u << (k * 8 + s);
            }
            for (int k = 0; k < 3; ++k)
            {
                double qlo = std::floor((c.box.lo[k] - (double)p[k]) / step[k]), qhi = std::ceil((c.box.hi[k] - (double)p[k]) / step[k]);
                qlo = std::min(std::max(qlo, 0.0), 255.0), qhi = std::min(std::max(qhi, 0.0), 255.0);
                const uint32_t word = 8 + 2 * (uint32_t)k + ((uint32_t)s >> 2), sh = 8 * ((uint32_t)s & 3u);
                w[word] |= (uint32_t)qlo << sh;
                w[word + 6] |= (uint32_t)qhi << sh;
            }
        }
        w[4] = (uint32_t)(index + 1);  // placeholder, fixed below
        w[5] = (uint32_t)out.tri_src.size();
        w[6] = tvalid | (imask << 24);
        for (uint32_t k = 0; k < kWideLeafMax; ++k)
            for (int s = 0; s < 8; ++s)
                if (leaf_tris[s].size() > k) out.tri_src.push_back(leaf_tris[s][k]);
        const uint32_t child_base = (uint32_t)(out.nodes.size() / kWideNodeWords);
        w[4]                      = child_base;
        uint32_t n_inner = 0;
        for (int s = 0; s < 8; ++s)
            if (imask & (1u << s)) ++n_inner;
        out.nodes.resize(out.nodes.size() + (size_t)n_inner * kWideNodeWords, 0u);  // invalidates w
        uint32_t rel = 0;
        for (int s = 0; s < 8; ++s)
            if (imask & (1u << s)) next_level.push_back({child_base + rel++, kids_in[(size_t)kid_at[s]]});
    }
};
}  // namespace

void build_wide_tree(const float* bnodes, uint32_t n, const float scene_lo[3], const float scene_hi[3], WideTree& out)
{
    out.nodes.clear(), out.tri_src.clear();
    out.depth = 0, out.top_nodes = 0;
    if (n == 0) return;
    double m = 0.0;
    for (int k = 0; k < 3; ++k)
        m = std::max({m, (double)scene_hi[k] - (double)scene_lo[k], std::fabs((double)scene_lo[k]), std::fabs((double)scene_hi[k])});
    Collapser c{bnodes, n, {}, (double)kWidePad * std::max(m, 1e-30), out};
    c.count_subtrees();
    out.nodes.assign(kWideNodeWords, 0u);
    out.tri_src.reserve(n);
    // breadth-first: `level` holds (wide node index, the binary subtree it stands for)
    std::vector<std::pair<uint32_t, Ref>> level, next;
    Ref root;
    root.node = n >= 2 ? 0 : ~0;
    root.count = n;
    for (int k = 0; k < 3; ++k)
    {
        // only used as a child box by the one-triangle scene: padded like the refit pads a triangle's box (bvh.hip k_refit)
        const double rp = 1e-5 * std::max(1.0, std::max(std::fabs((double)scene_lo[k]), std::fabs((double)scene_hi[k])));
        root.box.lo[k] = (double)scene_lo[k] - rp, root.box.hi[k] = (double)scene_hi[k] + rp;
    }
    level.push_back({0u, root});
    while (!level.empty())
    {
        ++out.depth;
        if (out.depth <= 3) out.top_nodes = (uint32_t)std::min<size_t>(out.nodes.size() / kWideNodeWords, kWideTopNodes);
        next.clear();
        for (const auto& item : level)
        {
            std::vector<Ref> start, kids;
            if (item.second.node >= 0)
                start = {c.ref_of((uint32_t)item.second.node, 0), c.ref_of((uint32_t)item.second.node, 1)};
            else
                start = {item.second};  // the one-triangle scene
            c.open(start, kids);
            c.emit(item.first, kids, next);
        }
        level.swap(next);
    }
}
}  // namespace cap

extern "C" int cap_host_wide_build(const float* nodes, uint32_t n, const float* scene_lo, const float* scene_hi, uint32_t* wide_nodes,
                                   uint32_t wide_capacity, uint32_t* tri_src, uint32_t* info)
{
    if ((n > 1 && !nodes) || !scene_lo || !scene_hi || !info || (n && (!wide_nodes || !tri_src))) return CAP_ERR_INVALID_ARG;
    cap::WideTree t;
    cap::build_wide_tree(nodes, n, scene_lo, scene_hi, t);
    const size_t count = t.nodes.size() / cap::kWideNodeWords;
    info[0] = (uint32_t)count, info[1] = t.depth, info[2] = t.top_nodes;
    if (count > wide_capacity) return CAP_ERR_INVALID_ARG;
    if (count) std::memcpy(wide_nodes, t.nodes.data(), sizeof(uint32_t) * t.nodes.size());
    for (size_t i = 0; i < t.tri_src.size(); ++i) tri_src[i] = t.tri_src[i];
    return CAP_OK;
}
