"""Diagnostic (needs a -DCAP_STAMPS build of the library): per-phase shader-clock sums of k_trace_shade<bounce >= 1> over one
step of the headline workload.  python tools/stamps.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capsaicin_amd import capi  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w, h, depth = 1920, 1080, 8
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
r = capi.Renderer(0)
r.upload_geometry(capi.Geometry(os.path.join(root, "assets", "cornell_box.obj")))
r.upload_bluenoise(capi.load_bluenoise())
r.build_bvh()
r.set_resolution(w, h)
r.set_camera(capi.cornell_camera(w, h))
r.render(0, spp, depth, 0)
r.sync()
out = (C.c_ulonglong * 8)()
L = capi.lib()
L.cap_debug_stamps(out, 1)
r.stats_reset()
r.render(0, spp, depth, capi.RENDER_STAGE_TIMERS)
s = r.stats()
L.cap_debug_stamps(out, 1)
names = ["queue entry wait", "triangle loop + winner record", "shade inputs + shading ALU", "append atomic", "stores issue", "-", "-", "-"]
chunks = s.rays_extension / 64.0
tot = float(sum(out))
print("closest stage %.2f ms, %d launches, %.0f chunks" % (s.ms_trace_closest, s.launches_trace_closest, chunks))
for n, v in zip(names, out):
    if v:
        print("  %-32s %8.0f cycles per chunk  %5.1f %%" % (n, v / chunks, 100.0 * v / tot))
print("  total per chunk %.0f cycles" % (tot / chunks))
