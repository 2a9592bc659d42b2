"""The binary-tree kernels behind the wide view: k_trace_closest_refill<32|64> and k_trace_any<32|64> are what
context.hip routes extension and shadow rays to when the compressed 8-wide view is deeper than the wide kernels' per-lane stacks
(21 levels: 262 k triangles are 8 deep, 16.8 M are 11), and under the A/B switch CAP_NO_WIDE8.  Nothing ran them in the driver's
suite until round 4.  Here:
 * in-process: the depth bound is lowered with cap_debug_set(CAP_DEBUG_WIDE_DEPTH_LIMIT) so that an ordinary scene's wide view
   counts as too deep; cap_debug_get(CAP_DEBUG_WIDE_IN_USE) says which kernels the render takes; all six planes, the accumulated
   image and the three ray counters must be bit-identical either way (the hit rule never looks at boxes); the 64-entry
   instantiations need a binary tree deeper than 32, which the 16.8 M-triangle hall has (tests/test_big_scene_gpu.py);
 * a child process with CAP_NO_WIDE8=1 renders the triangle soup and the small textured hall bit-identically to this process.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from capsaicin_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
PLANES = (capi.BUF_GBUFFER_GEO, capi.BUF_DIRECT, capi.BUF_ALBEDO, capi.BUF_NORMAL_DEPTH, capi.BUF_INDIRECT, capi.BUF_COMBINED)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def soup(n, seed=5):
    rs = np.random.RandomState(seed)
    c = rs.uniform(-1, 1, (n, 1, 3))
    tris = np.float32(c + rs.uniform(-0.15, 0.15, (n, 3, 3)))
    pos = tris.reshape(-1, 3)
    fn = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
    fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
    nrm = np.repeat(fn, 3, 0).astype(np.float32)
    return pos, nrm, np.zeros((len(pos), 2), np.float32), np.arange(len(pos), dtype=np.uint32), np.uint32([[len(pos), 0, len(pos), 0, 0, 0xFFFFFFFF, 0, 0]])


def soup_camera(w, h):
    return capi.camera_from_config(dict(position=(0.2, 0.1, 4.0), forward=(0.0, 0.0, -1.0), focal_length=0.03, sensor_x=0.036), w, h)


def progression(n=120):
    """triangles whose size and distance grow geometrically (tests/test_bvh_gpu.py): most of them in one Morton cell"""
    tris = []
    for k in range(n):
        s_ = 1.08 ** k
        x = 0.02 * s_
        tris.append([[x, -0.01 * s_, -0.002 * s_], [x + 0.01 * s_, 0.0, -0.002 * s_], [x, 0.01 * s_, -0.002 * s_]])
    pos = np.float32(tris).reshape(-1, 3)
    return (pos, np.tile(np.float32([0, 0, 1]), (len(pos), 1)), np.zeros((len(pos), 2), np.float32), np.arange(len(pos), dtype=np.uint32),
            np.uint32([[len(pos), 0, len(pos), 0, 0, 0xFFFFFFFF, 0, 0]]))


def hall(scale=0.15, tex=64):
    import make_sponza_class as gen
    return gen.arrays(scale, tex)


def hall_camera(w, h):
    import make_sponza_class as gen
    return capi.camera_from_config(dict(gen.camera(), sensor_x=0.036), w, h)


def render_all(r, frame, spp, depth, flags=0):
    """(planes of an AOV frame, accumulated image, ray counters) of the renderer as it stands"""
    r.accum_reset()
    r.stats_reset()
    r.render(frame, 1, depth, capi.RENDER_AOV | flags)
    planes = [r.readback(k) for k in PLANES]
    r.accum_reset()
    r.render(frame, spp, depth, flags)
    s = r.stats()
    assert s.guard_shade == 0 and s.guard_trace_any == 0 and s.guard_append == 0
    return planes, r.readback(capi.BUF_ACCUM_SUM), (s.rays_primary, s.rays_extension, s.rays_shadow)


def make(scene, bluenoise, w, h, build):
    r = capi.Renderer(0)
    r.set_bvh_build(build)
    if scene == "soup":
        r.upload_scene(*soup(3000))
        cam = soup_camera(w, h)
    elif scene == "progression":
        r.upload_scene(*progression())
        cam = capi.camera_from_config(dict(position=(40.0, 0.0, 120.0), forward=(0.0, 0.0, -1.0), focal_length=0.02, sensor_x=0.036), w, h)
    else:
        pos, nrm, uv, idx, meshes, texs = hall()
        r.upload_scene(pos, nrm, uv, idx, meshes)
        for i, t in enumerate(texs):
            r.upload_texture(i, t)
        cam = hall_camera(w, h)
    r.upload_bluenoise(bluenoise)
    info = r.build_bvh()
    r.set_resolution(w, h)
    r.set_camera(cam)
    r.set_traversal(1)
    return r, info


@pytest.mark.parametrize("scene,build", [("soup", capi.Renderer.BVH_BUILD_AUTO), ("soup", capi.Renderer.BVH_BUILD_LBVH), ("hall", capi.Renderer.BVH_BUILD_AUTO),
                                         ("hall", capi.Renderer.BVH_BUILD_SAH), ("progression", capi.Renderer.BVH_BUILD_SAH),
                                         ("progression", capi.Renderer.BVH_BUILD_LBVH)])
def test_too_deep_a_wide_view_takes_the_binary_kernels(native_lib, bluenoise, scene, build):
    w, h, spp, depth = 160, 96, 3, 4
    r, info = make(scene, bluenoise, w, h, build)
    wide_nodes, wide_depth, _ = r.bvh_wide_info()
    assert wide_depth >= 3 and r.debug_get(capi.Renderer.DEBUG_WIDE_IN_USE) == 1
    assert info.stack_entries == (32 if info.max_depth <= 32 else 64)  # (the 64-entry instantiations: tests/test_big_scene_gpu.py, depth 36)
    want = render_all(r, 2, spp, depth)
    r.debug_set(capi.Renderer.DEBUG_WIDE_DEPTH_LIMIT, 2)  # "the stacks end at depth 2": this view no longer fits
    assert r.debug_get(capi.Renderer.DEBUG_WIDE_IN_USE) == 0
    got = render_all(r, 2, spp, depth)
    for a, b in zip(want[0], got[0]):
        assert np.array_equal(bits(a), bits(b))
    assert np.array_equal(bits(want[1]), bits(got[1])) and want[2] == got[2]
    # ... and with the EXT shading model (its shadow rays go through k_trace_any<.., true>)
    if scene == "soup":
        mats = np.zeros((1, 12), np.float32)
        mats[0, 0:3], mats[0, 3], mats[0, 4:7], mats[0, 8:11] = 0.6, 0.4, 0.2, (0.5, 0.4, 0.3)
        r.upload_materials(mats)
        got_ext = render_all(r, 1, spp, depth, capi.RENDER_EXT_MATERIALS)
        r.debug_set(capi.Renderer.DEBUG_WIDE_DEPTH_LIMIT, 0)
        assert r.debug_get(capi.Renderer.DEBUG_WIDE_IN_USE) == 1
        want_ext = render_all(r, 1, spp, depth, capi.RENDER_EXT_MATERIALS)
        for a, b in zip(want_ext[0], got_ext[0]):
            assert np.array_equal(bits(a), bits(b))
        assert np.array_equal(bits(want_ext[1]), bits(got_ext[1])) and want_ext[2] == got_ext[2]
    r.close()


CHILD = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import test_fallback_kernels_gpu as T
from capsaicin_amd import capi
bn = capi.load_bluenoise()
out = {{}}
for scene in ("soup", "hall"):
    r, info = T.make(scene, bn, 160, 96, capi.Renderer.BVH_BUILD_AUTO)
    assert r.debug_get(capi.Renderer.DEBUG_WIDE_IN_USE) == 0, "CAP_NO_WIDE8 must put the rays on the binary kernels"
    planes, acc, rays = T.render_all(r, 2, 3, 4)
    out[scene + "_acc"] = acc
    out[scene + "_rays"] = np.uint64(rays)
    for i, p in enumerate(planes):
        out["%s_p%d" % (scene, i)] = p
    r.close()
np.savez({path!r}, **out)
"""


def test_no_wide8_child_process_is_bit_identical(native_lib, bluenoise, tmp_path):
    path = str(tmp_path / "child.npz")
    env = dict(os.environ, CAP_NO_WIDE8="1")
    p = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, tests=os.path.join(ROOT, "tests"), path=path)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    child = np.load(path)
    for scene in ("soup", "hall"):
        r, info = make(scene, bluenoise, 160, 96, capi.Renderer.BVH_BUILD_AUTO)
        assert r.debug_get(capi.Renderer.DEBUG_WIDE_IN_USE) == 1
        planes, acc, rays = render_all(r, 2, 3, 4)
        assert np.array_equal(bits(acc), bits(child[scene + "_acc"])) and tuple(int(x) for x in child[scene + "_rays"]) == rays
        for i, pl in enumerate(planes):
            assert np.array_equal(bits(pl), bits(child["%s_p%d" % (scene, i)])), (scene, i)
        r.close()


CHILD_DENSE = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import test_fallback_kernels_gpu as T
from capsaicin_amd import capi
bn = capi.load_bluenoise()
out = {{}}
for scene in ("soup", "hall"):
    for shard in ((0, 1), (1, 3)):
        r, info = T.make(scene, bn, 160, 96, capi.Renderer.BVH_BUILD_AUTO)
        r.set_shard(*shard)
        planes, acc, rays = T.render_all(r, 2, 3, 4)
        key = "%s_%d_%d" % (scene, shard[0], shard[1])
        out[key + "_acc"] = acc
        out[key + "_rays"] = np.uint64(rays)
        for i, p in enumerate(planes):
            out["%s_p%d" % (key, i)] = p
        r.close()
np.savez({path!r}, **out)
"""


def test_dense_scene_kernels_forced_on_small_scenes(native_lib, bluenoise, tmp_path):
    """The two kernels dense scenes switch to -- camera rays as an identity queue through k_trace_closest8 (from one triangle per
    pixel on) and shadow rays through k_trace_any8_refill (from 512 MiB of tree on) -- forced on small scenes in a child process
    (CAP_PRIMARY_WIDE=1 CAP_ANY_REFILL=1), unsharded and on shard 1 of 3: all six planes, the accumulated image and the ray counters
    are those of this process, which takes the packet walk and the per-chunk kernel.  (At their own sizes:
    tests/test_big_scene_gpu.py.)"""
    path = str(tmp_path / "dense.npz")
    env = dict(os.environ, CAP_PRIMARY_WIDE="1", CAP_ANY_REFILL="1")
    p = subprocess.run([sys.executable, "-c", CHILD_DENSE.format(root=ROOT, tests=os.path.join(ROOT, "tests"), path=path)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    child = np.load(path)
    for scene in ("soup", "hall"):
        for shard in ((0, 1), (1, 3)):
            r, info = make(scene, bluenoise, 160, 96, capi.Renderer.BVH_BUILD_AUTO)
            r.set_shard(*shard)
            planes, acc, rays = render_all(r, 2, 3, 4)
            key = "%s_%d_%d" % (scene, shard[0], shard[1])
            assert np.array_equal(bits(acc), bits(child[key + "_acc"])) and tuple(int(x) for x in child[key + "_rays"]) == rays, key
            for i, pl in enumerate(planes):
                assert np.array_equal(bits(pl), bits(child["%s_p%d" % (key, i)])), (key, i)
            r.close()


def test_switch_table_flips_paths_in_one_process(native_lib, bluenoise):
    """The A/B switches are one table per context (cap_debug_set(CAP_DEBUG_SWITCH_BASE + index); the environment only fills it at
    cap_ctx_create): the paths the child-process tests above force through the environment, flipped here on ONE context between renders
    -- binary-tree kernels instead of the 8-wide view, camera rays through k_trace_closest8, the lane-refill any-hit kernel, one batch
    lane, the other builders' parameters -- every combination bit-identical, counters included."""
    r, info = make("hall", bluenoise, 160, 96, capi.Renderer.BVH_BUILD_AUTO)
    want = render_all(r, 2, 3, 4)
    assert r.debug_get(capi.Renderer.DEBUG_WIDE_IN_USE) == 1
    with pytest.raises(capi.CapError):
        r.debug_switch("CAP_NO_SUCH_SWITCH", 1)

    def same(what):
        got = render_all(r, 2, 3, 4)
        assert got[2] == want[2], what
        assert np.array_equal(bits(got[1]), bits(want[1])), what
        for a, b in zip(got[0], want[0]):
            assert np.array_equal(bits(a), bits(b)), what

    r.debug_switch("CAP_NO_WIDE8", 1)
    assert r.debug_get(capi.Renderer.DEBUG_WIDE_IN_USE) == 0
    same("binary-tree kernels")
    r.debug_switch("CAP_NO_WIDE8", None)
    assert r.debug_get(capi.Renderer.DEBUG_WIDE_IN_USE) == 1
    r.debug_switch("CAP_PRIMARY_WIDE", 1)
    r.debug_switch("CAP_ANY_REFILL", 1)
    same("dense-scene kernels")
    r.debug_switch("CAP_NO_TWO_LANES", 1)
    same("dense-scene kernels, one lane")
    assert r.debug_get(capi.Renderer.DEBUG_LANES_USED) == 1
    r.debug_switch("CAP_PRIMARY_WIDE", 0)
    r.debug_switch("CAP_ANY_REFILL", 0)
    r.debug_switch("CAP_NO_PACKET", 1)
    same("no packet walk")
    for name in ("CAP_NO_TWO_LANES", "CAP_PRIMARY_WIDE", "CAP_ANY_REFILL", "CAP_NO_PACKET"):
        r.debug_switch(name, None)
    # the builders' switches: another leaf size and search window, the clustering alone -- other trees, the same image
    for name, value in (("CAP_SAHDEV_LEAF", 64), ("CAP_PLOC_RADIUS", 8), ("CAP_AUTO_SAH_TRIANGLES", 1 << 30)):
        r.debug_switch(name, value)
        r.build_bvh()
        same("%s = %d" % (name, value))
    r.close()
