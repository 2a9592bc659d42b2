"""Times the reconstruction chain (cap_post_frame) at a given resolution on cuda:0: GPU ms per frame of the chain alone.
python tools/time_post.py [width height frames [fast]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capsaicin_amd import capi  # noqa: E402


def main():
    args = sys.argv[1:4]
    w, h, n = (int(a) for a in (args + ["1920", "1080", "20"][len(args):]))
    fast = len(sys.argv) > 4 and sys.argv[4] == "fast"
    r = capi.Renderer(0)
    r.upload_geometry(capi.Geometry(os.path.join(os.path.dirname(capi.LIB_PATH), "..", "assets", "cornell_box.obj")))
    r.upload_bluenoise(capi.load_bluenoise())
    r.build_bvh()
    r.set_resolution(w, h)
    cam = capi.cornell_camera(w, h)
    r.set_camera(cam)
    s = capi.PostSettings(fast_weights=1 if fast else 0)
    r.render(0, 1, 2, capi.RENDER_AOV)
    for f in range(3):
        r.post_frame(s, f, cam)
    r.sync()
    t0 = time.perf_counter()
    for f in range(3, 3 + n):
        r.post_frame(s, f, cam)
    r.sync()
    ms = (time.perf_counter() - t0) * 1e3 / n
    # algorithmic bytes per pixel: 11 float4 reads + 8 float4 writes of distinct images per frame
    gbs = w * h * 16 * 19 / (ms * 1e-3) / 1e9
    print("post chain %dx%d (%s weights): %.3f ms/frame, %.1f GB/s algorithmic" % (w, h, "fast" if fast else "exact", ms, gbs))


if __name__ == "__main__":
    main()
