// wide_builder.h — host-side collapse of the binary traversal tree (device LBVH or host SAH build, same node layout) into the
// compressed 8-wide view of cap_wide.h.  One-off work per scene, like the build itself: the reference asks its driver for
// PREFER_FAST_TRACE and builds once (blas_system.cpp:42-47, tlas_system.cpp:111-121).
#pragma once

#include <cstdint>
#include <vector>

namespace cap
{
struct WideTree
{
    std::vector<uint32_t> nodes;    // kWideNodeWords per node, breadth-first (root = node 0, top levels = a prefix of the array)
    std::vector<uint32_t> tri_src;  // wide-order triangle record i = leaf-order (sorted) triangle tri_src[i] of the binary tree
    uint32_t              depth = 0;  // wide nodes on the longest root-to-leaf path (the traversal stack needs depth - 1 entries)
    uint32_t              top_nodes = 0;  // leading nodes that are the root, its children and grandchildren, capped at kWideTopNodes
};

// bnodes: n - 1 binary nodes, 16 floats each (cap_device.h "BVH node": child boxes lo0 hi0 lo1 hi1, then child0 child1 tchild0
// tchild1 as int bits; a negative child is ~(leaf-order triangle index)); may be null when n < 2.
// scene_lo / scene_hi: bounds of all triangles (the only box there is when n == 1).
void build_wide_tree(const float* bnodes, uint32_t n, const float scene_lo[3], const float scene_hi[3], WideTree& out);
}  // namespace cap
