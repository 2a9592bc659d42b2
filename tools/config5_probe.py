"""Why is config 5's share slower per ray than config 3?  Steady-state timings of the Cornell render by shading model, resolution,
shard and depth.  python tools/config5_probe.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from capsaicin_amd import capi
def run(w, h, spp, depth, shard, ext=True, batch=0):
    fl = capi.RENDER_EXT_MATERIALS if ext else 0
    r = bench.make_cornell(0, None, w, h, ext=ext, shard=shard)
    if batch: r.set_batch_paths(batch)
    r.render(0, spp, depth, fl)  # every buffer at its final size before anything is timed
    dt, st = bench.timed(r, 0, spp, depth, fl, 2)
    rays = (st.rays_primary + st.rays_extension + st.rays_shadow) / 2
    _, sp = bench.timed(r, 0, spp, depth, fl | capi.RENDER_STAGE_TIMERS, 1)
    print("%s %dx%d spp %d depth %d shard %s: %.1f ms, %.2f Grays/s, rays/path %.2f | primary %.1f closest %.1f (%d launches) any %.1f resolve %.1f" %
          ("EXT" if ext else "REF", w, h, spp, depth, shard, dt * 1e3, rays / dt / 1e9, 2 * rays / st.rays_primary, sp.ms_primary, sp.ms_trace_closest,
           sp.launches_trace_closest, sp.ms_trace_any, sp.ms_resolve))
    r.close()
for ext in (False, True):
    run(4096, 4096, 16, 8, (0, 1), ext)
    run(4096, 4096, 32, 8, (0, 2), ext)
    run(4096, 4096, 64, 8, (0, 4), ext)
    run(4096, 4096, 128, 8, (0, 8), ext)
    run(4096, 4096, 128, 8, (3, 8), ext)
    run(4000, 4000, 128, 8, (0, 8), ext)
