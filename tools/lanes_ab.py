"""Step time with the context's own stream (not a torch stream): CAP_NO_TWO_LANES=1 python tools/lanes_ab.py  vs  python tools/lanes_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from capsaicin_amd import capi

def mk(scene, shard):
    r = capi.Renderer(0)
    if scene == "sponza":
        cam = bench.load_sponza_class(r)
    else:
        r.upload_geometry(capi.Geometry(os.path.join(ROOT, "assets", "cornell_box.obj"))); cam = capi.cornell_camera(1920, 1080)
    r.upload_bluenoise(capi.load_bluenoise()); r.build_bvh(); r.set_resolution(1920, 1080); r.set_camera(cam); r.set_shard(*shard)
    return r


def main():
    for scene, spp in (("cornell", 64), ("sponza", 32)):
        for shard in ((0, 1), (0, 8)):
            r = mk(scene, shard)
            def step():
                r.accum_reset(); r.render(0, spp, 8, 0); r.sync()
            step(); step()
            t0 = time.perf_counter()
            for _ in range(5): step()
            print("%-8s shard %d/%d  %s  %.2f ms per step" % (scene, shard[0], shard[1], "one lane " if os.environ.get("CAP_NO_TWO_LANES") else "two lanes", (time.perf_counter() - t0) / 5 * 1e3), flush=True)
            r.close()


if __name__ == "__main__":
    main()
