"""A/B of the tree builders on the 262 k-triangle scene: stage split per step (16 spp, 1080p, depth 8) for CAP_BVH_BUILD_LBVH
(on-device Morton build) and CAP_BVH_BUILD_SAH (host binned SAH), both followed by the 8-wide collapse.  Run through gpurun."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from capsaicin_amd import capi  # noqa: E402

MODES = ((1, "LBVH (device)"), (2, "SAH (host)"), (3, "PLOC (device)"), (4, "SAH (device)"))
for mode, name in [mn for mn in MODES if len(sys.argv) < 2 or str(mn[0]) in sys.argv[1:]]:
    r = capi.Renderer(0)
    r.set_bvh_build(mode)
    cam = bench.load_sponza_class(r)
    r.upload_bluenoise(capi.load_bluenoise())
    bi = r.build_bvh()
    r.set_resolution(bench.WIDTH, bench.HEIGHT)
    r.set_camera(cam)
    r.render(0, 16, bench.DEPTH, capi.RENDER_STAGE_TIMERS)
    r.sync()
    r.stats_reset()
    for _ in range(2):
        r.accum_reset()
        r.render(0, 16, bench.DEPTH, capi.RENDER_STAGE_TIMERS)
    s = r.stats()
    wnodes, _, wdepth, _ = r.bvh_wide_readback()
    bn, _ = r.bvh_readback()
    ext = lambda lo, hi: (hi - lo)[:, 0] * (hi - lo)[:, 1] + (hi - lo)[:, 1] * (hi - lo)[:, 2] + (hi - lo)[:, 2] * (hi - lo)[:, 0]
    a0, a1 = ext(bn[:, 0:3], bn[:, 3:6]), ext(bn[:, 6:9], bn[:, 9:12])
    kid = bn[:, 12:14].copy().view(np.int32)
    root = ext(np.minimum(bn[:1, 0:3], bn[:1, 6:9]), np.maximum(bn[:1, 3:6], bn[:1, 9:12]))[0]
    inner = (a0[kid[:, 0] >= 0].sum() + a1[kid[:, 1] >= 0].sum()) / root + 1.0  # expected node visits of a random long ray
    leaf = (a0[kid[:, 0] < 0].sum() + a1[kid[:, 1] < 0].sum()) / root           # expected triangle tests
    print("   binary tree: expected node visits %.1f, triangle tests %.1f" % (inner, leaf))
    # the same two sums over the compressed 8-wide view (cap_wide.h layout)
    w = wnodes.astype(np.uint32)
    stp = np.stack([w[:, 3], w[:, 7] & np.uint32(0xffff0000), (w[:, 7] << np.uint32(16))], 1).view(np.float32).astype(np.float64)
    sh = (np.arange(8, dtype=np.uint32) & 3) * 8
    q = np.stack([(w[:, 8 + 2 * a_ + (np.arange(8) >> 2)] >> sh) & 0xff for a_ in range(6)], 1).astype(np.float64)  # (N, 6, 8)
    d = (q[:, 3:6, :] - q[:, 0:3, :]) * stp[:, :, None]
    area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]  # (N, 8)
    imask = (w[:, 6] >> 24)[:, None] >> np.arange(8)[None, :] & 1
    tv = w[:, 6] & 0xffffff
    ntri = sum(((tv >> (8 * k))[:, None] >> np.arange(8)[None, :]) & 1 for k in range(3))
    print("   8-wide view: expected node visits %.1f, triangle tests %.1f, children per node %.2f, triangles per leaf child %.2f" %
          (1.0 + (area * imask).sum() / root, (area * ntri).sum() / root, ((imask + (ntri > 0)).sum() / len(w)), ntri.sum() / max(1, (ntri > 0).sum())))
    print("%-14s build %.1f ms depth %d, wide %d nodes depth %d | ms/step: total %.2f primary %.2f closest %.2f any %.2f shade %.2f" %
          (name, bi.build_ms, bi.max_depth, len(wnodes), wdepth, s.ms_total / 2, s.ms_primary / 2, s.ms_trace_closest / 2, s.ms_trace_any / 2, s.ms_shade / 2))
    r.close()
