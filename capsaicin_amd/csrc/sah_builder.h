// sah_builder.h — host-side binned-SAH build of the traversal tree (the "prefer fast trace" counterpart of the on-device LBVH:
// the reference asks the driver for D3D12_RAYTRACING_ACCELERATION_STRUCTURE_BUILD_FLAG_PREFER_FAST_TRACE and builds once,
// blas_system.cpp:42-47, tlas_system.cpp:111-121).  Produces exactly the device layout of bvh.hip: a complete binary tree of
// n - 1 internal nodes, 16 floats each (cap_device.h "BVH node"), and the leaf order of the triangles.
#pragma once

#include <cstdint>
#include <vector>

namespace cap
{
struct HostTree
{
    std::vector<float>    nodes;  // 16 floats per internal node, root = node 0
    std::vector<uint32_t> order;  // order[i] = global triangle id of sorted (leaf) position i
    uint32_t              depth = 0;  // internal nodes on the longest root-to-leaf path
};

// tri_box: n x 8 floats (lo.xyz, -, hi.xyz, -) as written by k_tri_setup (unpadded).  leaf_max / count_shift: the traversal
// leaf coding of cap_device.h (kLeafMax, kLeafCountShift).
void build_sah_tree(const float* tri_box, uint32_t n, int leaf_max, uint32_t count_shift, HostTree& out);
}  // namespace cap
