"""A/B of the tree builders on the 262 k-triangle scene: stage split per step (16 spp, 1080p, depth 8) for CAP_BVH_BUILD_LBVH
(on-device Morton build) and CAP_BVH_BUILD_SAH (host binned SAH), both followed by the 8-wide collapse.  Run through gpurun."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from capsaicin_amd import capi  # noqa: E402

for mode, name in ((1, "LBVH (device)"), (2, "SAH (host)")):
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
    print("%-14s build %.1f ms depth %d | ms/step: total %.2f primary %.2f closest %.2f any %.2f shade %.2f" %
          (name, bi.build_ms, bi.max_depth, s.ms_total / 2, s.ms_primary / 2, s.ms_trace_closest / 2, s.ms_trace_any / 2, s.ms_shade / 2))
    r.close()
