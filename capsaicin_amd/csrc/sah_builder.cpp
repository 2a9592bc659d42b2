// sah_builder.cpp — top-down binned surface-area-heuristic build on the host (see sah_builder.h).
// The traversal result does not depend on the tree shape (closest hit = min t, ties to the lower triangle id; DESIGN.md
// "Intersection contract"), so this builder only has to produce conservative boxes: every triangle box is padded exactly as
// the device refit pads it (bvh.hip k_refit).
#include "sah_builder.h"

#include "../../include/capsaicin_scene.h"
#include "cap_leaf.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace cap
{
namespace
{
struct Box
{
    float lo[3], hi[3];
    void  reset()
    {
        for (int k = 0; k < 3; ++k) lo[k] = INFINITY, hi[k] = -INFINITY;
    }
    void grow(const Box& b)
    {
        for (int k = 0; k < 3; ++k) lo[k] = std::fmin(lo[k], b.lo[k]), hi[k] = std::fmax(hi[k], b.hi[k]);
    }
    void grow(const float p[3])
    {
        for (int k = 0; k < 3; ++k) lo[k] = std::fmin(lo[k], p[k]), hi[k] = std::fmax(hi[k], p[k]);
    }
    float half_area() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kBins = 32;

struct Builder
{
    const std::vector<Box>&   box;       // padded triangle boxes, by global id
    const std::vector<float>& centroid;  // 3 per triangle
    std::vector<uint32_t>&    order;
    std::vector<float>&       nodes;
    int                       leaf_max;
    uint32_t                  count_shift;
    uint32_t                  next_node = 0;
    uint32_t                  max_depth = 0;

    uint32_t leaf_code(uint32_t first, uint32_t count) const { return ~(first | ((count - 1u) << count_shift)); }

    // builds the subtree over order[b, e) (e - b >= 2) into a fresh node; returns its index and its box
    uint32_t build(uint32_t b, uint32_t e, uint32_t depth, Box& out_box)
    {
        const uint32_t node = next_node++;
        max_depth           = std::max(max_depth, depth);
        // very unbalanced SAH splits (one triangle against the rest, over and over) would make the tree deeper than the
        // traversal stacks allow: from depth 36 on the split is the median, which bounds the depth by 36 + log2(range)
        const uint32_t mid  = split(b, e, depth < 36);
        Box            cb[2];
        uint32_t       child[2], tchild[2];
        const uint32_t rb[2] = {b, mid}, re[2] = {mid, e};
        for (int s = 0; s < 2; ++s)
        {
            const uint32_t cnt = re[s] - rb[s];
            if (cnt == 1)
            {
                cb[s]    = box[order[rb[s]]];
                child[s] = ~rb[s];
            }
            else
                child[s] = build(rb[s], re[s], depth + 1, cb[s]);
            tchild[s] = (int)cnt <= leaf_max ? leaf_code(rb[s], cnt) : child[s];
        }
        float* q = nodes.data() + 16 * (size_t)node;
        for (int s = 0; s < 2; ++s)
            for (int k = 0; k < 3; ++k) q[6 * s + k] = cb[s].lo[k], q[6 * s + 3 + k] = cb[s].hi[k];
        std::memcpy(q + 12, &child[0], 4), std::memcpy(q + 13, &child[1], 4);
        std::memcpy(q + 14, &tchild[0], 4), std::memcpy(q + 15, &tchild[1], 4);
        out_box = cb[0];
        out_box.grow(cb[1]);
        return node;
    }

    // partitions order[b, e) and returns the split position (b < mid < e)
    uint32_t split(uint32_t b, uint32_t e, bool use_sah)
    {
        const uint32_t n = e - b;
        Box            cbox;
        cbox.reset();
        for (uint32_t i = b; i < e; ++i) cbox.grow(&centroid[3 * (size_t)order[i]]);
        int   best_axis = -1, best_bin = 0;
        float best_cost = INFINITY;
        if (n > 4 && use_sah)
        {
            for (int axis = 0; axis < 3; ++axis)
            {
                const float ext = cbox.hi[axis] - cbox.lo[axis];
                if (!(ext > 0.0f)) continue;
                Box      bb[kBins];
                uint32_t cnt[kBins] = {0};
                for (auto& x : bb) x.reset();
                const float scale = (float)kBins / ext;
                for (uint32_t i = b; i < e; ++i)
                {
                    const uint32_t g = order[i];
                    int            k = (int)((centroid[3 * (size_t)g + axis] - cbox.lo[axis]) * scale);
                    k                = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                    bb[k].grow(box[g]);
                    ++cnt[k];
                }
                float    right_area[kBins];
                uint32_t right_cnt[kBins];
                Box      acc;
                acc.reset();
                uint32_t c = 0;
                for (int k = kBins - 1; k > 0; --k)
                {
                    acc.grow(bb[k]);
                    c += cnt[k];
                    right_area[k] = c ? acc.half_area() : 0.0f, right_cnt[k] = c;
                }
                acc.reset();
                c = 0;
                for (int k = 0; k < kBins - 1; ++k)
                {
                    acc.grow(bb[k]);
                    c += cnt[k];
                    if (c == 0 || right_cnt[k + 1] == 0) continue;
                    const float cost = acc.half_area() * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
                    if (cost < best_cost) best_cost = cost, best_axis = axis, best_bin = k;
                }
            }
        }
        if (best_axis >= 0)
        {
            const float lo = cbox.lo[best_axis], scale = (float)kBins / (cbox.hi[best_axis] - cbox.lo[best_axis]);
            const auto  mid = std::partition(order.begin() + b, order.begin() + e, [&](uint32_t g) {
                int k = (int)((centroid[3 * (size_t)g + best_axis] - lo) * scale);
                k     = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                return k <= best_bin;
            });
            const uint32_t m = (uint32_t)(mid - order.begin());
            if (m > b && m < e) return m;
        }
        // small or degenerate ranges: median along the widest centroid axis (ties by id keep it deterministic)
        int axis = 0;
        for (int k = 1; k < 3; ++k)
            if (cbox.hi[k] - cbox.lo[k] > cbox.hi[axis] - cbox.lo[axis]) axis = k;
        const uint32_t m = b + n / 2;
        std::nth_element(order.begin() + b, order.begin() + m, order.begin() + e, [&](uint32_t x, uint32_t y) {
            const float cx = centroid[3 * (size_t)x + axis], cy = centroid[3 * (size_t)y + axis];
            return cx < cy || (cx == cy && x < y);
        });
        return m;
    }
};
}  // namespace

void build_sah_tree(const float* tri_box, uint32_t n, int leaf_max, uint32_t count_shift, HostTree& out)
{
    out.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) out.order[i] = i;
    out.nodes.assign(16 * (size_t)(n > 1 ? n - 1 : 1), 0.0f);
    out.depth = 0;
    if (n < 2) return;
    std::vector<Box>   box(n);
    std::vector<float> centroid(3 * (size_t)n);
    for (uint32_t g = 0; g < n; ++g)
    {
        const float* t = tri_box + 8 * (size_t)g;
        for (int k = 0; k < 3; ++k)
        {
            const float lo = t[k], hi = t[4 + k];
            centroid[3 * (size_t)g + k] = (lo + hi) * 0.5f;
            // the device refit's padding (bvh.hip k_refit): the box must contain every point the fp32 triangle test can report
            const float pad = 1e-5f * std::fmax(1.0f, std::fmax(std::fabs(lo), std::fabs(hi)));
            box[g].lo[k] = lo - pad, box[g].hi[k] = hi + pad;
        }
    }
    Builder b{box, centroid, out.order, out.nodes, leaf_max, count_shift};
    Box     root;
    b.build(0, n, 1, root);
    out.depth = b.max_depth;
}
}  // namespace cap

extern "C" int cap_host_sah_build(const float* tri_boxes, uint32_t n, float* nodes, uint32_t* order, uint32_t* depth)
{
    if ((!tri_boxes && n) || !order || (n > 1 && !nodes)) return CAP_ERR_INVALID_ARG;
    cap::HostTree t;
    cap::build_sah_tree(tri_boxes, n, cap::kLeafMax, cap::kLeafCountShift, t);
    for (uint32_t i = 0; i < n; ++i) order[i] = t.order[i];
    if (n > 1) std::memcpy(nodes, t.nodes.data(), sizeof(float) * 16 * (size_t)(n - 1));
    if (depth) *depth = t.depth;
    return CAP_OK;
}
