"""tests/golden/cornell_frames.npz (tools/make_golden.py): frames of the ray passes and of the feedback + reconstruction loop,
computed once by the oracle and committed.  CPU: the oracle still reproduces them bit for bit (drift guard).  GPU: the HIP path
reproduces them through the C ABI."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 40, 24
PLANES = ("gbuffer_geo", "direct", "albedo", "normal_depth", "indirect", "combined")


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "cornell_frames.npz"))


def _ocam(O, dx=0.0):
    return O.make_camera((-0.01 + dx, 0.995, 3.4), (0.0, 0.0, -1.0), (-1.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.036,
                         float(np.float32(0.036) * (np.float32(H) / np.float32(W))), 0.035)


def test_oracle_reproduces_golden(golden, bluenoise, cornell_path):
    from oracle import cap_oracle as O
    from oracle import obj_oracle
    g = obj_oracle.load_geometry(cornell_path)
    sc = O.Scene(g["positions"], g["normals"], g["texcoords"], g["indices"], g["meshes"])
    ref = sc.render_frame(_ocam(O), bluenoise, W, H, 3, 2, threads=4)
    for k in PLANES:
        assert np.array_equal(ref[k].view(np.uint32), golden["f3_d2_" + k]), k
    assert tuple(int(x) for x in golden["f3_d2_rays"]) == ref["rays"]
    bvh = sc.render_frame(_ocam(O), bluenoise, W, H, 3, 2, flags=O.FLAG_USE_BVH, threads=4)  # CPU BVH == brute force
    for k in PLANES:
        assert np.array_equal(bvh[k].view(np.uint32), golden["f3_d2_" + k]), k
    chain = O.PostChain(W, H)
    prev, pnd, hist = _ocam(O), np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.float32)
    for f in range(4):
        cam = _ocam(O, 0.02 * max(0, f - 1))
        r = sc.render_frame(cam, bluenoise, W, H, f, 2, threads=4, feedback=(prev, pnd, hist))
        o = chain.frame(O.PostSettings(), f, cam, prev, r)
        assert np.array_equal(r["indirect"].view(np.uint32), golden["loop_f%d_indirect" % f]), f
        assert np.array_equal(o.view(np.uint32), golden["loop_f%d_output" % f]), f
        prev, pnd, hist = cam, r["normal_depth"], o


@pytest.mark.gpu
@pytest.mark.parametrize("traversal", [1, 2])
def test_hip_reproduces_golden(golden, native_lib, bluenoise, cornell_path, traversal):
    from capsaicin_amd import capi
    kinds = dict(gbuffer_geo=capi.BUF_GBUFFER_GEO, direct=capi.BUF_DIRECT, albedo=capi.BUF_ALBEDO, normal_depth=capi.BUF_NORMAL_DEPTH,
                 indirect=capi.BUF_INDIRECT, combined=capi.BUF_COMBINED)

    def cam_at(dx):
        c = capi.cornell_camera(W, H)
        c.position[0] = np.float32(-0.01 + dx)
        return c

    r = capi.Renderer(0)
    r.upload_geometry(capi.Geometry(cornell_path))
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_traversal(traversal)
    r.set_resolution(W, H)
    r.set_camera(cam_at(0.0))
    r.stats_reset()
    r.render(3, 1, 2, capi.RENDER_AOV)
    for k in PLANES:
        assert np.array_equal(r.readback(kinds[k]).view(np.uint32), golden["f3_d2_" + k]), k
    s = r.stats()
    assert (s.rays_primary, s.rays_extension, s.rays_shadow) == tuple(int(x) for x in golden["f3_d2_rays"])
    prev = cam_at(0.0)
    for f in range(4):
        cam = cam_at(0.02 * max(0, f - 1))
        r.set_camera(cam)
        r.set_prev_camera(prev)
        r.render(f, 1, 2, capi.RENDER_AOV | capi.RENDER_GBUFFER_FEEDBACK)
        assert np.array_equal(r.readback(capi.BUF_INDIRECT).view(np.uint32), golden["loop_f%d_indirect" % f]), f
        r.post_frame(capi.PostSettings(), f, prev)
        assert np.array_equal(r.post_readback().view(np.uint32), golden["loop_f%d_output" % f]), f
        prev = cam
    r.close()
