// cap_leaf.h — the traversal-leaf coding shared by the device code (cap_device.h) and the host-side tree builder.
#pragma once

#include <stdint.h>

namespace cap
{
// Tree traversal leaves hold up to kLeafMax consecutive sorted triangles, coded ~(first | (count - 1) << kLeafCountShift)
#ifndef CAP_LEAF_MAX
#define CAP_LEAF_MAX 2  // 262 k-triangle scene, ms per step: 1: 31.0, 2: 28.6, 3: 29.2, 4: 30.4, 8: 34.7
#endif
constexpr int      kLeafMax        = CAP_LEAF_MAX;  // <= 8 (3 bits)
constexpr uint32_t kLeafCountShift = 27;
constexpr uint32_t kLeafFirstMask  = (1u << kLeafCountShift) - 1u;  // < 134 M triangles
}  // namespace cap
