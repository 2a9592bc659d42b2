"""Camera rays: packet walk on the binary tree vs one lane per ray on the wide view (context.hip primary_wide), per scene density.
    CAP_PRIMARY_WIDE=0|1 python tools/primary_ab.py SCALE [SPP]     -> stage split of a stage-timed render of the hall at that scale"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from capsaicin_amd import capi  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r, bi = bench.make_hall(0, None, scale=scale)
r.render(0, spp, bench.DEPTH, 0)
_, s = bench.timed(r, 0, spp, bench.DEPTH, capi.RENDER_STAGE_TIMERS, 2)
print("scale %g (%d triangles, %.2f per pixel) CAP_PRIMARY_WIDE=%s CAP_ANY_REFILL=%s: primary %.2f closest %.2f any %.2f shade %.2f total %.2f ms per %d spp" %
      (scale, bi.triangle_count, bi.triangle_count / (bench.WIDTH * bench.HEIGHT), os.environ.get("CAP_PRIMARY_WIDE", "auto"), os.environ.get("CAP_ANY_REFILL", "auto"), s.ms_primary / 2,
       s.ms_trace_closest / 2, s.ms_trace_any / 2, s.ms_shade / 2, s.ms_total / 2, spp))
r.close()
