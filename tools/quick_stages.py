"""Stage split of the headline workload (Cornell 1080p, 64 spp, depth 8) on cuda:0 — a quick A/B probe for kernel experiments.
python tools/quick_stages.py [steps] [ext]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capsaicin_amd import capi  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ext = len(sys.argv) > 2 and sys.argv[2] == "ext"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    w, h, spp, depth = 1920, 1080, 64, 8
    r = capi.Renderer(0)
    geo = capi.Geometry(os.path.join(root, "assets", "cornell_box.obj"))
    r.upload_geometry(geo)
    if ext:
        import shutil
        import tempfile
        d = tempfile.mkdtemp()
        txt = open(os.path.join(root, "assets", "cornell_box.obj")).read().replace("mtllib cornellbox.mtl", "mtllib cornell_box.mtl")
        open(os.path.join(d, "c.obj"), "w").write(txt)
        shutil.copy(os.path.join(root, "assets", "cornell_box.mtl"), os.path.join(d, "cornell_box.mtl"))
        g2 = capi.Geometry(os.path.join(d, "c.obj"))
        r.upload_materials(g2.materials())
    r.upload_bluenoise(capi.load_bluenoise())
    r.build_bvh()
    r.set_resolution(w, h)
    r.set_camera(capi.cornell_camera(w, h))
    flags = capi.RENDER_STAGE_TIMERS | (capi.RENDER_EXT_MATERIALS if ext else 0)
    r.render(0, spp, depth, flags)
    r.sync()
    r.stats_reset()
    for _ in range(steps):
        r.accum_reset()
        r.render(0, spp, depth, flags)
    s = r.stats()
    rays = (s.rays_primary + s.rays_extension + s.rays_shadow) / steps
    print("ms/step: total %.2f  primary %.2f  closest %.2f  any %.2f  shade %.2f  resolve %.2f   | %.2f Grays/s" %
          (s.ms_total / steps, s.ms_primary / steps, s.ms_trace_closest / steps, s.ms_trace_any / steps, s.ms_shade / steps,
           s.ms_resolve / steps, rays / (s.ms_total / steps) / 1e6))


if __name__ == "__main__":
    main()
