"""Host-side SAH tree builder (capsaicin_amd/csrc/sah_builder.cpp) through the C ABI, no GPU: the tree has the device layout
(complete binary tree of n - 1 nodes, every triangle in exactly one leaf), its boxes contain their content with the refit's
padding, the traversal pointers collapse small subtrees into ranges of consecutive sorted triangles, and adversarial inputs
stay within the traversal stack."""
import numpy as np
import pytest

from capsaicin_amd import capi

LEAF_SHIFT, LEAF_MASK = 27, (1 << 27) - 1


def walk(nodes, order, lo, hi):
    n = len(order)
    assert sorted(order.tolist()) == list(range(n))
    if n < 2:
        return 0
    child = nodes[:, 12:14].copy().view(np.int32)
    tchild = nodes[:, 14:16].copy().view(np.int32)
    seen_nodes, seen_leaves = np.zeros(n - 1, bool), np.zeros(n, bool)
    pad = lambda a: 1e-5 * np.maximum(1.0, np.abs(a))  # noqa: E731

    def content(c):
        """(sorted positions covered, box) of child pointer c"""
        if c < 0:
            g = order[~c]
            return [~c], lo[g] - pad(lo[g]) - 1e-30, hi[g] + pad(hi[g]) + 1e-30
        q = nodes[c]
        return None, np.minimum(q[0:3], q[6:9]), np.maximum(q[3:6], q[9:12])

    ranges = {}
    depth = 0
    stack = [(0, 1)]
    post = []
    while stack:
        i, d = stack.pop()
        assert not seen_nodes[i]
        seen_nodes[i] = True
        depth = max(depth, d)
        post.append(i)
        for slot in range(2):
            c = int(child[i, slot])
            blo, bhi = (nodes[i, 0:3], nodes[i, 3:6]) if slot == 0 else (nodes[i, 6:9], nodes[i, 9:12])
            _, clo, chi = content(c)
            assert np.all(blo <= clo + 1e-6) and np.all(bhi >= chi - 1e-6), (i, slot)
            if c < 0:
                assert not seen_leaves[~c]
                seen_leaves[~c] = True
            else:
                stack.append((c, d + 1))
    assert seen_nodes.all() and seen_leaves.all()
    # sorted-position ranges bottom-up, then the traversal pointers
    for i in reversed(post):
        r = []
        for slot in range(2):
            c = int(child[i, slot])
            r += [~c] if c < 0 else ranges[c]
        assert r == list(range(r[0], r[0] + len(r))), "a subtree covers consecutive sorted positions"
        ranges[i] = r
    for i in post:
        for slot in range(2):
            c, t = int(child[i, slot]), int(tchild[i, slot])
            r = [~c] if c < 0 else ranges[c]
            if len(r) <= 2:
                code = ~t
                assert t < 0 and (code & LEAF_MASK) == r[0] and (code >> LEAF_SHIFT) + 1 == len(r)
            else:
                assert t == c
    return depth


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 64, 1000, 20000])
def test_random_boxes(native_lib, n):
    rs = np.random.RandomState(n)
    c = rs.uniform(-10, 10, (n, 3)).astype(np.float32)
    e = rs.uniform(0, 0.5, (n, 3)).astype(np.float32)
    nodes, order, depth = capi.host_sah_build(c - e, c + e)
    assert walk(nodes, order, c - e, c + e) == depth
    if n >= 1000:
        assert depth <= 3 * int(np.ceil(np.log2(n)))  # SAH on uniform input stays close to balanced


def test_identical_and_degenerate_boxes(native_lib):
    lo = np.zeros((300, 3), np.float32)
    nodes, order, depth = capi.host_sah_build(lo, lo)  # all centroids equal: median splits by id
    assert walk(nodes, order, lo, lo) == depth and depth <= 10
    lo = np.repeat(np.float32([[0, 0, 0], [1, 0, 0]]), 50, axis=0)
    nodes, order, depth = capi.host_sah_build(lo, lo + np.float32(0.0))
    assert walk(nodes, order, lo, lo) == depth


def test_geometric_progression_depth_is_bounded(native_lib):
    n = 400
    s = (1.05 ** np.arange(n)).astype(np.float32)
    lo = np.stack([0.02 * s, -0.01 * s, -0.002 * s], 1).astype(np.float32)
    hi = np.stack([0.03 * s, 0.01 * s, -0.002 * s], 1).astype(np.float32)
    nodes, order, depth = capi.host_sah_build(lo, hi)
    assert walk(nodes, order, lo, hi) == depth
    assert depth <= 36 + int(np.ceil(np.log2(n)))  # the builder's stated bound; the traversal stacks hold 64
