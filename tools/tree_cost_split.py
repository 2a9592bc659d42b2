"""Where does the clustering tree's surface-area cost exceed the host SAH tree's -- at the top of the tree or near the leaves?
For each builder on the hall at a given scale: the SAH cost sum (child box area / root area over inner children = expected node visits of a
random long ray) split by the size of the child's subtree (triangles below it).  python tools/tree_cost_split.py [scale]   (through gpurun)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from capsaicin_amd import capi  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
EDGES = [1, 8, 64, 512, 4096, 32768, 262144, 1 << 30]
for mode, name in ((3, "PLOC (device)"), (4, "SAH (device)"), (2, "SAH (host)")):
    r = capi.Renderer(0)
    r.set_bvh_build(mode)
    bench.load_sponza_class(r, scale=scale)
    bi = r.build_bvh()
    bn, _ = r.bvh_readback()
    r.close()
    n = len(bn)
    ext = lambda lo, hi: (hi - lo)[:, 0] * (hi - lo)[:, 1] + (hi - lo)[:, 1] * (hi - lo)[:, 2] + (hi - lo)[:, 2] * (hi - lo)[:, 0]
    a = np.stack([ext(bn[:, 0:3], bn[:, 3:6]), ext(bn[:, 6:9], bn[:, 9:12])], 1)
    kid = bn[:, 12:14].copy().view(np.int32)
    root = ext(np.minimum(bn[:1, 0:3], bn[:1, 6:9]), np.maximum(bn[:1, 3:6], bn[:1, 9:12]))[0]
    # triangles below every node: children are visited before parents in a reverse BFS order
    order, head = [0], 0
    while head < len(order):
        v = order[head]
        head += 1
        for s in (0, 1):
            if kid[v, s] >= 0:
                order.append(int(kid[v, s]))
    cnt = np.zeros(n, np.int64)
    leaf_cnt = lambda link: (((~link) >> 27) & 0x1f) + 1  # cap_leaf.h: ~(first | (count - 1) << kLeafCountShift); only used if the build emits multi-triangle leaves
    for v in reversed(order):
        c = 0
        for s in (0, 1):
            c += cnt[kid[v, s]] if kid[v, s] >= 0 else 1
        cnt[v] = c
    below = np.where(kid >= 0, cnt[np.maximum(kid, 0)], 1)
    cost = a / root
    out = []
    for lo, hi in zip(EDGES[:-1], EDGES[1:]):
        m = (below >= lo) & (below < hi) & (kid >= 0)
        out.append("%d..: %.2f" % (lo, cost[m].sum()))
    leaf = cost[kid < 0].sum()
    print("%-14s scale %g  %d triangles, depth %d | inner-child cost by subtree size: %s | total %.2f, leaf children %.2f" %
          (name, scale, bi.triangle_count, bi.max_depth, "  ".join(out), cost[kid >= 0].sum(), leaf), flush=True)
