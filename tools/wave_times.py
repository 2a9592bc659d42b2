"""Diagnostic (needs a -DCAP_STAMPS build): start/end time of every wave of one bounce >= 1 launch of the fused kernel."""
import ctypes as C
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capsaicin_amd import capi  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w, h, depth = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 1
r = capi.Renderer(0)
r.upload_geometry(capi.Geometry(os.path.join(root, "assets", "cornell_box.obj")))
r.upload_bluenoise(capi.load_bluenoise())
r.build_bvh()
r.set_resolution(w, h)
r.set_camera(capi.cornell_camera(w, h))
r.render(0, 4, depth, 0)
r.sync()
r.render(0, 4, depth, 0)  # one batch of 4 frames: the last fused launch is bounce `depth`
r.sync()
buf = np.zeros(2 * 16384, np.uint64)
capi.lib().cap_debug_stamps(buf.ctypes.data_as(C.c_void_p), 2)
t = buf.reshape(-1, 2).astype(np.int64)
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0  # microseconds (100 MHz)
print("waves %d  kernel span %.1f us" % (len(t), end.max()))
print("start  percentiles 0/50/90/99/100: %s" % np.percentile(start, [0, 50, 90, 99, 100]).round(1))
print("end    percentiles 0/1/10/50/90/100: %s" % np.percentile(end, [0, 1, 10, 50, 90, 100]).round(1))
print("life   percentiles 0/10/50/90/100: %s  mean %.1f" % (np.percentile(end - start, [0, 10, 50, 90, 100]).round(1), (end - start).mean()))
blk = (end - start).reshape(-1, 4).mean(1)
print("per-workgroup mean life: first 8 %s ... last 8 %s" % (blk[:8].round(1), blk[-8:].round(1)))
# per class (wave index mod 64): when its last wave left -- the spread is what work stealing between classes could recover
n = (len(t) // 64) * 64
cls_end = end[:n].reshape(-1, 64).max(0)
print("per-class last exit: min %.1f  median %.1f  max %.1f us; mean wave life / span = %.3f" % (
    cls_end.min(), np.median(cls_end), cls_end.max(), (end - start).mean() / end.max()))
