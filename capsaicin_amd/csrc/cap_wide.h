// cap_wide.h — the compressed 8-wide view of the traversal tree: layout shared by the host-side collapse (wide_builder.cpp) and
// the gfx950 kernels (trace8.hip).
//
// Why: in round 1 the per-lane traversal of incoherent rays was bound by what each lane pulls through the CU's texture-address
// path (profiles/r01_tree_path.txt: TA busy 64 % of the closest-hit launch, vector ALU 38 %; with this view the vector ALU is the
// first limiter and that path the second: docs/experiments.md (60)).  The binary
// tree costs 64 B per two child boxes, its 4-wide view 112 B per four; this node holds EIGHT child boxes in 80 B — child planes
// quantised to 8 bits on a per-node power-of-two grid (Ylitie, Karras, Laine 2017, "Efficient incoherent ray traversal on GPUs
// through compressed wide BVHs": the idea; layout, child indexing and traversal order code below are this build's own).
//
// The hit rule of the intersection contract (minimum t, ties to the lower triangle id) does not depend on the visiting order or on
// which conservative boxes are used, so images are bit-identical to every other traversal of the build.
//
// Node, 80 B = 5 x 16 B:
//   w0..w2   p.xyz      float: the node's grid origin (<= every child's low corner)
//   w3       step.x     float, a power of two: the grid step per axis (w7: step.y's upper 16 bits | step.z's upper 16 bits >> 16)
//   w4       child_base: node index of the first inner child; inner children are contiguous in slot order
//   w5       tri_base:   index of the node's first triangle record; a leaf child holds 1..kWideLeafMax triangles
//   w6       tvalid | imask << 24:  imask bit s = slot s is an inner child;  tvalid bit (k * 8 + s) = slot s is a leaf child with
//            more than k triangles.  Triangle (k, s) is record tri_base + popcount(tvalid & ((1 << (k * 8 + s)) - 1)).
//   w8..w19  quantised planes, one byte per slot, slots 0..3 in the first word: lo.x[8] lo.y[8] lo.z[8] hi.x[8] hi.y[8] hi.z[8]
//            child plane = p + q * step, low planes rounded down, high planes rounded up (exact in the reals: the build works in
//            double), after padding the child boxes by kWidePad * (scene size) — see wide_builder.cpp for the error budget of
//            the fp32 slab arithmetic the padding pays for.
// Slot assignment: slot s stands for the direction ((s & 1) ? + : -, (s & 2) ? + : -, (s & 4) ? + : -); children are placed to
// maximise the sum of <child centre - node centre, direction(slot)>.  A ray visits hit slots in descending (s ^ octinv),
// octinv = 7 - (sign bits of its direction): front to back along its octant's diagonal.
#pragma once

#include <stdint.h>

namespace cap
{
constexpr uint32_t kWideNodeWords = 20;     // 80 B
// Words between two nodes in device memory.  20: packed (a node straddles 128-byte lines: 1.5 on average).  32: every node in its own
// 128-byte line -- what an L2 miss fetches (docs/experiments.md (58)).
#ifndef CAP_WIDE_STRIDE_WORDS
#define CAP_WIDE_STRIDE_WORDS 20
#endif
constexpr uint32_t kWideNodeStride = CAP_WIDE_STRIDE_WORDS;
static_assert(kWideNodeStride >= kWideNodeWords && kWideNodeStride % 4u == 0u, "node stride: whole 16-byte pieces");
constexpr uint32_t kWideLeafMax   = 3;      // triangles per leaf child (k = 0..2: the three byte lanes of tvalid)
constexpr float    kWidePad       = 4e-6f;  // child boxes grow by this times max(scene diagonal extent, largest |coordinate|)
constexpr uint32_t kWideTopNodes  = 73;     // nodes 0 .. kWideTopNodes-1 (breadth-first: root, its children, their children at most)
                                            // are what a workgroup may keep in LDS
}  // namespace cap
