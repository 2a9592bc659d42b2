"""Sanity run at the sizes of BASELINE configs 3 and 5 (3840x2160 depth 8; 4096x4096 depth 16) with a few spp: finite output, full
sample counts, no guard hits, identical bits under a different batching.  Run through gpurun."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from capsaicin_amd import capi
r = capi.Renderer(0)
r.upload_geometry(capi.Geometry(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "cornell_box.obj")))
r.upload_bluenoise(capi.load_bluenoise())
r.build_bvh()
for (w, h, spp, D) in ((4096, 4096, 3, 16), (3840, 2160, 4, 8)):
    r.set_resolution(w, h); r.set_camera(capi.cornell_camera(w, h))
    r.accum_reset(); r.stats_reset()
    t0 = time.perf_counter(); r.render(0, spp, D); r.sync(); dt = time.perf_counter() - t0
    a = r.readback(capi.BUF_ACCUM_SUM); s = r.stats()
    rays = s.rays_primary + s.rays_extension + s.rays_shadow
    assert np.isfinite(a).all() and (a[..., 3] == spp).all() and s.rays_primary == spp * w * h, "bad"
    assert s.guard_shade == 0 and s.guard_trace_any == 0
    r.set_batch_paths(w * h); r.accum_reset(); r.render(0, spp, D)
    b = r.readback(capi.BUF_ACCUM_SUM)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "batching changed bits"
    r.set_batch_paths(0)
    print("%dx%d spp %d depth %d: %.1f ms, %.1f Grays/s, mean %.4f" % (w, h, spp, D, dt * 1e3, rays / dt / 1e9, float(a[..., :3].mean() / spp)))
