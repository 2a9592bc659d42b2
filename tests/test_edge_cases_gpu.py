"""Edge cases of the ray passes against the oracle, bit for bit: degenerate geometry, rays through shared edges and vertices,
tiny and odd resolutions, more frames than frame slots, frame counters at the uint32 wrap, depth 0."""
import numpy as np
import pytest

from capsaicin_amd import capi

pytestmark = pytest.mark.gpu

PLANES = (("gbuffer_geo", capi.BUF_GBUFFER_GEO), ("direct", capi.BUF_DIRECT), ("albedo", capi.BUF_ALBEDO),
          ("normal_depth", capi.BUF_NORMAL_DEPTH), ("indirect", capi.BUF_INDIRECT), ("combined", capi.BUF_COMBINED))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def camera(w, h, pos=(0.0, 0.0, 4.0)):
    cam = capi.CameraData()
    cam.position[:] = pos
    cam.forward[:] = (0, 0, -1)
    cam.right[:] = (-1, 0, 0)
    cam.up[:] = (0, 1, 0)
    cam.focal_length = 0.03
    cam.sensor_size[0] = 0.036
    cam.sensor_size[1] = np.float32(0.036) * (np.float32(h) / np.float32(w))
    return cam


def ocam(O, cam):
    return O.make_camera(tuple(cam.position), tuple(cam.forward), tuple(cam.right), tuple(cam.up), cam.sensor_size[0], cam.sensor_size[1],
                         cam.focal_length)


def scene_arrays(tris):
    """tris: (n, 3, 3) -> unindexed GeometryStorage with face normals (degenerate faces get (0,0,1))."""
    tris = np.float32(tris)
    pos = tris.reshape(-1, 3)
    fn = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
    ln = np.linalg.norm(fn, axis=1, keepdims=True)
    fn = np.where(ln > 0, fn / np.maximum(ln, 1e-30), np.float32([0, 0, 1]))
    nrm = np.repeat(fn, 3, axis=0).astype(np.float32)
    uv = np.zeros((len(pos), 2), np.float32)
    idx = np.arange(len(pos), dtype=np.uint32)
    meshes = np.uint32([[len(pos), 0, len(pos), 0, 0, 0xFFFFFFFF, 0, 0]])
    return pos, nrm, uv, idx, meshes


def check_modes(arrays, bluenoise, w, h, frame, D, cam):
    from oracle import cap_oracle as O
    r = capi.Renderer(0)
    r.upload_scene(*arrays)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(w, h)
    r.set_camera(cam)
    sc = O.Scene(*arrays)
    ref = sc.render_frame(ocam(O, cam), bluenoise, w, h, frame, D, threads=4)
    for mode in (1, 2):
        r.set_traversal(mode)
        r.stats_reset()
        r.render(frame, 1, D, capi.RENDER_AOV)
        for name, kind in PLANES:
            got = r.readback(kind)
            nbad = int((bits(got) != bits(ref[name])).any(-1).sum())
            assert nbad == 0, "%s (traversal %d): %d pixels differ" % (name, mode, nbad)
        s = r.stats()
        assert (s.rays_primary, s.rays_extension, s.rays_shadow) == ref["rays"]
        assert s.guard_shade == 0 and s.guard_trace_any == 0
    r.close()
    return ref


def test_degenerate_and_duplicate_triangles(native_lib, bluenoise):
    """Zero-area triangles (collapsed to a segment and to a point), exact duplicates (equal t: the lower id wins) and a
    sliver, next to ordinary geometry."""
    quad = [[[-1, -1, 0], [1, -1, 0], [1, 1, 0]], [[-1, -1, 0], [1, 1, 0], [-1, 1, 0]]]
    tris = quad + quad  # exact duplicates
    tris += [[[0, 0, 1], [0.5, 0.5, 1], [1, 1, 1]]]  # collinear
    tris += [[[0.3, 0.3, 0.5]] * 3]  # a point
    tris += [[[-1, 0.2, 0.5], [1, 0.2, 0.5], [1, 0.2000001, 0.5]]]  # sliver
    tris += [[[-0.5, -0.5, 0.7], [0.0, -0.5, 0.7], [-0.5, 0.0, 0.7]]]
    ref = check_modes(scene_arrays(tris), bluenoise, 64, 48, 3, 3, camera(64, 48))
    geo = ref["gbuffer_geo"].view(np.uint32)
    hit = geo[..., 3] != 0xFFFFFFFF
    assert hit.any() and set(np.unique(geo[..., 3][hit]).tolist()) <= {0, 1, 6, 7}  # duplicates 2, 3 never win; 4, 5 are never hit


def test_rays_through_shared_edges_and_vertices(native_lib, bluenoise):
    """An axis-aligned fan of four triangles around the origin, camera on the axis with frame 1's jitter (0.25, 0.667) and an even
    resolution: rays pass close to and exactly along shared edges; every pixel must pick the same triangle as the oracle."""
    c = [0, 0, 0]
    ring = [[1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0]]
    tris = [[c, ring[k], ring[(k + 1) % 4]] for k in range(4)]
    tris += [[[2 * p for p in ring[k]], [2 * p for p in ring[(k + 1) % 4]], [0, 0, -1]] for k in range(4)]
    for w, h, frame in ((32, 32, 0), (33, 31, 1), (8, 8, 2)):
        check_modes(scene_arrays(tris), bluenoise, w, h, frame, 2, camera(w, h, pos=(0.0, 0.0, 3.0)))


@pytest.mark.parametrize("w,h", [(1, 1), (9, 3), (8, 8), (7, 130)])
def test_tiny_and_odd_resolutions(native_lib, bluenoise, cornell_path, w, h):
    from oracle import cap_oracle as O
    geo = capi.Geometry(cornell_path)
    arrays = (geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes)
    check_modes(arrays, bluenoise, w, h, 5, 2, capi.cornell_camera(w, h))


def test_more_frames_than_slots_and_counter_wrap(native_lib, bluenoise, cornell_path):
    """70 frames of a 16x16 image: more than the 64 frame slots of a batch, so the call is split; and frame counters around
    2^32 (frame * 25 + bounce and frame % 4096 wrap exactly as the reference's uint arithmetic does)."""
    from oracle import cap_oracle as O
    geo = capi.Geometry(cornell_path)
    w = h = 16
    r = capi.Renderer(0)
    r.upload_geometry(geo)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(w, h)
    cam = capi.cornell_camera(w, h)
    r.set_camera(cam)
    sc = O.Scene(geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes)
    acc, rays = sc.render_accumulate(ocam(O, cam), bluenoise, w, h, 0, 70, 3, threads=8)
    r.accum_reset()
    r.stats_reset()
    r.render(0, 70, 3, 0)
    got = r.readback(capi.BUF_ACCUM_SUM)
    assert np.array_equal(bits(got[..., :3]), bits(acc[..., :3])) and np.all(got[..., 3] == 70.0)
    s = r.stats()
    assert (s.rays_primary, s.rays_extension, s.rays_shadow) == rays
    for frame in (2**32 - 1, 2**32 - 2, 171798692):  # 171798692 * 25 overflows uint32
        r.render(frame, 1, 2, capi.RENDER_AOV)
        ref = sc.render_frame(ocam(O, cam), bluenoise, w, h, frame, 2, threads=4)
        for name, kind in PLANES:
            assert np.array_equal(bits(r.readback(kind)), bits(ref[name])), (frame, name)
    r.close()


def test_depth_zero_and_max_depth(native_lib, bluenoise, cornell_path):
    geo = capi.Geometry(cornell_path)
    arrays = (geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes)
    check_modes(arrays, bluenoise, 40, 30, 9, 0, capi.cornell_camera(40, 30))
    check_modes(arrays, bluenoise, 24, 16, 9, 40, capi.cornell_camera(24, 16))
    r = capi.Renderer(0)
    r.upload_scene(*arrays)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(8, 8)
    r.set_camera(capi.cornell_camera(8, 8))
    with pytest.raises(capi.CapError, match="num_bounces"):
        r.render(0, 1, 256, 0)
    r.close()


def _open_top_scene():
    """A floor and two low walls under an open sky: the occluder the producer-side probe tests first (the pair farthest along the
    light, which points up: lighting.h:20-33) shadows almost nothing, so nearly every shadow ray of bounces >= 1 survives the probe
    and travels through its wave's 128-entry ring (kernels.hip trace_ring)."""
    def quad(a, b, c, d):
        return [[a, b, c], [a, c, d]]
    tris = quad([-2, 0, 2], [2, 0, 2], [2, 0, -2], [-2, 0, -2])                    # floor, normal +y
    tris += quad([-2, 0, -2], [2, 0, -2], [2, 0.6, -2], [-2, 0.6, -2])              # back wall, normal +z
    tris += quad([-2, 0, 2], [-2, 0, -2], [-2, 0.6, -2], [-2, 0.6, 2])              # left wall, normal +x
    tris += quad([0.2, 0, 0.2], [0.8, 0, 0.2], [0.8, 0.9, 0.2], [0.2, 0.9, 0.2])   # a panel, normal +z
    return scene_arrays(tris)


def _ring_camera(w, h):
    cam = capi.CameraData()
    f = np.float64([0.0, -0.55, -1.0])
    f /= np.linalg.norm(f)
    right = -np.cross(f, (0, 1, 0))
    right /= np.linalg.norm(right)
    cam.position[:] = (0.0, 2.2, 3.6)
    cam.forward[:] = f
    cam.right[:] = right
    cam.up[:] = np.cross(f, right)
    cam.focal_length = 0.03
    cam.sensor_size[0] = 0.036
    cam.sensor_size[1] = np.float32(0.036) * (np.float32(h) / np.float32(w))
    return cam


_RING_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/tests")
from capsaicin_amd import capi
import test_edge_cases_gpu as T
w, h, frames, depth = (int(x) for x in sys.argv[3:7])
bn = np.fromfile(sys.argv[1] + "/assets/bluenoise256.rgba", np.uint8).reshape(256, 256, 4)
r = capi.Renderer(0)
r.upload_scene(*T._open_top_scene()); r.upload_bluenoise(bn); r.build_bvh(); r.set_resolution(w, h); r.set_camera(T._ring_camera(w, h))
r.set_traversal(2); r.render(0, frames, depth, 0)
s = r.stats()
np.savez(sys.argv[2], acc=r.readback(capi.BUF_ACCUM_SUM), rays=np.uint64([s.rays_primary, s.rays_extension, s.rays_shadow, s.shadow_entries]))
"""


def test_wave_ring_open_top_scene(native_lib, bluenoise, tmp_path):
    """The per-wave ring as the ONLY route of the shadow rays: 32 frames of 1024x768 leave every persistent wave
    (at most 256 CUs x 8 workgroups x 4) several hundred survivors per launch, so each 128-entry ring wraps several times.  Bit-exact against the oracle, and bit- and
    counter-identical to a process that runs without the ring (CAP_NO_WAVE_RING: the any-hit launch traces the same entries)."""
    import os
    import subprocess
    import sys
    from oracle import cap_oracle as O
    w, h, frames, depth = 1024, 768, 32, 3
    arrays, cam = _open_top_scene(), _ring_camera(w, h)
    r = capi.Renderer(0)
    r.upload_scene(*arrays)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(w, h)
    r.set_camera(cam)
    r.set_traversal(2)
    r.render(0, frames, depth, 0)
    got, s = r.readback(capi.BUF_ACCUM_SUM), r.stats()
    r.close()
    acc, rays = O.Scene(*arrays).render_accumulate(ocam(O, cam), bluenoise, w, h, 0, frames, depth, threads=8)
    assert np.array_equal(bits(got[..., :3]), bits(acc[..., :3])) and np.all(got[..., 3] == frames)
    assert (s.rays_primary, s.rays_extension, s.rays_shadow) == rays and s.guard_shade == 0 and s.guard_trace_any == 0
    # the scene does what it is for: the probe answers (almost) nothing, so the survivors -- counted in shadow_entries -- are most
    # of the shadow rays, and there are enough of them to wrap every wave's ring
    ring_entries = s.shadow_entries - s.shadow_entries_bounce0
    assert s.shadow_entries > 0.8 * s.rays_shadow and ring_entries > 4 * 256 * 8 * 4 * 128
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "noring.npz")
    env = dict(os.environ, CAP_NO_WAVE_RING="1")
    subprocess.run([sys.executable, "-c", _RING_CHILD, root, out, str(w), str(h), str(frames), str(depth)], check=True, env=env, timeout=300)
    z = np.load(out)
    assert np.array_equal(bits(z["acc"]), bits(got))
    assert tuple(int(x) for x in z["rays"]) == (s.rays_primary, s.rays_extension, s.rays_shadow, s.shadow_entries)
