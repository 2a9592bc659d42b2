"""Step time of one render against the same frames cut over two / four contexts on one device: python tools/contexts_ab.py"""
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
            a, b, c4 = mk(scene, shard), mk(scene, shard), [mk(scene, shard) for _ in range(2)]
            def one():
                a.render(0, spp, 8, 0); a.sync()
            def two():
                a.render(0, spp // 2, 8, 0); b.render(spp // 2, spp // 2, 8, 0); a.sync(); b.sync()
            def four():
                q = spp // 4
                for i, r in enumerate([a, b] + c4): r.render(i * q, q, 8, 0)
                for r in [a, b] + c4: r.sync()
            for name, fn in (("one context", one), ("two contexts", two), ("four contexts", four)):
                fn(); fn()
                t0 = time.perf_counter()
                for _ in range(4): fn()
                print("%-8s shard %d/%d  %-14s %.2f ms per step" % (scene, shard[0], shard[1], name, (time.perf_counter() - t0) / 4 * 1e3), flush=True)
            for r in [a, b] + c4: r.close()


if __name__ == "__main__":
    main()
