"""Stage split of the procedural hall (tools/make_sponza_class.py) on the tree path, one whole batch per lane-less render -- the quick
A/B probe for traversal-kernel experiments on the cache-resident (scale 1: 262 k triangles, 32 spp) and the HBM-bound (scale 8:
16.8 M triangles, 8 spp) scene.
    python tools/hall_stages.py [scale] [spp] [build]              one process, prints one line
    python tools/hall_stages.py ab <scale> <spp> "A=1 B=2" "C=3"   one child process per environment set (statics cache the switches),
                                                                   the plain environment first and last
build: 0 AUTO (device clustering), 1 device Morton hierarchy, 2 host SAH."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(scale, spp, build):
    import bench
    from capsaicin_amd import capi
    r = capi.Renderer(0)
    cam = bench.load_sponza_class(r, scale=scale)
    r.upload_bluenoise(capi.load_bluenoise())
    if build:
        r.set_bvh_build(build)
    bi = r.build_bvh()
    r.set_resolution(bench.WIDTH, bench.HEIGHT)
    r.set_camera(cam)
    r.set_batch_paths(spp * r.tile_buffer_floats() // 4)
    depth = bench.DEPTH
    r.render(0, spp, depth, capi.RENDER_STAGE_TIMERS)
    r.sync()
    out = []
    for _ in range(2):
        r.stats_reset()
        t0 = time.perf_counter()
        r.accum_reset()
        r.render(0, 2 * spp, depth, capi.RENDER_STAGE_TIMERS)
        r.sync()
        wall = (time.perf_counter() - t0) / 2 * 1e3
        s = r.stats()
        rays = (s.rays_primary + s.rays_extension + s.rays_shadow) / 2
        out.append("total %.2f  primary %.2f  closest %.2f  any %.2f  shade %.2f  resolve %.2f | %.2f Grays/s wall %.2f" %
                   (s.ms_total / 2, s.ms_primary / 2, s.ms_trace_closest / 2, s.ms_trace_any / 2, s.ms_shade / 2, s.ms_resolve / 2,
                    rays / (s.ms_total / 2) / 1e6, wall))
        if s.guard_shade or s.guard_trace_any or s.guard_append:
            out.append("GUARDS FIRED %d %d %d" % (s.guard_shade, s.guard_trace_any, s.guard_append))
    # the plain two-lane step too (what bench.py times)
    r.set_batch_paths(0)
    r.render(0, spp, depth, 0)
    r.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        r.accum_reset()
        r.render(0, spp, depth, 0)
    r.sync()
    out.append("plain step %.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
    print("%d tris build %.1f ms depth %d | ms per %d spp: %s" % (bi.triangle_count, bi.build_ms, bi.max_depth, spp, "  ||  ".join(out)), flush=True)
    r.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "ab":
        scale, spp = sys.argv[2], sys.argv[3]
        sets = [""] + sys.argv[4:] + [""]
        for s in sets:
            env = dict(os.environ)
            for kv in s.split():
                k, _, v = kv.partition("=")
                env[k] = v
            print("== %s" % (s or "(plain)"), flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), scale, spp, env.get("HALL_BUILD", "0")], env=env, timeout=400)
        return
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if scale < 4 else 8)
    build = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    one(scale, spp, build)


if __name__ == "__main__":
    main()
