import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
from capsaicin_amd import capi
from oracle import cap_oracle as O
import test_post_gpu as T
bn = capi.load_bluenoise()
w, h, D = 150, 101, 2
geo = capi.Geometry(ROOT + "/assets/cornell_box.obj")
for settings in (dict(),) if os.environ.get("FAST_ERR_ONE") else (dict(), dict(use_variance=0), dict(eaw5=0, gather_luma_sigma=1.0)):
    r = capi.Renderer(0); r.upload_geometry(geo); r.upload_bluenoise(bn); r.build_bvh(); r.set_resolution(w, h)
    sc = O.Scene(geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes)
    chain = O.PostChain(w, h)
    base = capi.cornell_camera(w, h)
    cams = [base] * 4 + [T.moved(base, 0.02 * k, 0.01 * k, -0.03 * k) for k in range(1, 4)] + [T.moved(base, 0.06, 0.03, -0.09)] * 2
    gs, os_ = capi.PostSettings(fast_weights=1, **settings), O.PostSettings(**settings)
    prev = cams[0]
    for f, cam in enumerate(cams):
        r.set_camera(cam); r.render(f, 1, D, capi.RENDER_AOV); r.post_frame(gs, f, prev)
        got = r.post_readback()
        ref = sc.render_frame(T.ocam_of(O, cam), bn, w, h, f, D, threads=8)
        want = chain.frame(os_, f, T.ocam_of(O, cam), T.ocam_of(O, prev), ref)
        rel = np.abs(got.astype(np.float64) - want)[..., :3] / (np.abs(want[..., :3]) + 1e-3)
        print(settings, "frame", f, "rel err max %.2e  p99.9 %.2e  p99 %.2e  median %.2e  mean |want| %.3f" % (rel.max(), np.percentile(rel, 99.9), np.percentile(rel, 99), np.median(rel), np.abs(want[..., :3]).mean()))
        prev = cam
    r.close()
