"""EXT shading model (GGX + emissive triangles + next-event estimation; BASELINE configs 2, 3, 5).  It has
no reference counterpart: the specification is the oracle's shade_pixel_ext (DESIGN.md "EXT shading model") and the bar is
bit-exact agreement of the independent HIP implementation with it."""
import os
import shutil

import numpy as np
import pytest

from capsaicin_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def cornell_with_materials(tmp_path):
    """The Cornell box with its MTL actually resolved (the shipped OBJ names a missing file): Kd colours, emissive ceiling lamp."""
    txt = open(os.path.join(ROOT, "assets", "cornell_box.obj")).read().replace("mtllib cornellbox.mtl", "mtllib cornell_box.mtl")
    (tmp_path / "c.obj").write_text(txt)
    shutil.copy(os.path.join(ROOT, "assets", "cornell_box.mtl"), tmp_path / "cornell_box.mtl")
    geo = capi.Geometry(str(tmp_path / "c.obj"))
    mats = geo.materials()
    assert geo.material_count == 8 and mats[0, 8] == 36.0  # light: Ke 36 33 24
    return geo, mats


@pytest.mark.parametrize("ggx", [False, True])
def test_ext_cornell_parity(native_lib, bluenoise, tmp_path, ggx):
    from oracle import cap_oracle as O
    geo, mats = cornell_with_materials(tmp_path)
    if ggx:  # BASELINE configs[1]/[4]: "Lambert+GGX": make the two boxes and the back wall glossy
        for m, (rough, ks) in capi.scene_config()["cornell_ggx"].items():
            mats[int(m), 3] = rough
            mats[int(m), 4:7] = ks
    w, h, D = 96, 80, 5
    r = capi.Renderer(0)
    r.upload_geometry(geo)
    r.upload_materials(mats)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    cam = capi.cornell_camera(w, h)
    r.set_resolution(w, h)
    r.set_camera(cam)
    sc = O.Scene(geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes, materials=mats)
    ocam = O.make_camera(tuple(cam.position), tuple(cam.forward), tuple(cam.right), tuple(cam.up), cam.sensor_size[0], cam.sensor_size[1], cam.focal_length)
    ref = sc.render_frame(ocam, bluenoise, w, h, 11, D, flags=O.FLAG_EXT_MATERIALS)
    assert ref["rays"][2] > 0 and float(ref["combined"][..., :3].max()) > 1.0  # NEE reaches the lamp; the lamp is visible
    for mode in (1, 2):  # LBVH + stack kernels, exhaustive fused kernels
        r.set_traversal(mode)
        r.accum_reset()
        r.stats_reset()
        r.render(11, 1, D, capi.RENDER_AOV | capi.RENDER_EXT_MATERIALS)
        for name, kind in (("gbuffer_geo", capi.BUF_GBUFFER_GEO), ("direct", capi.BUF_DIRECT), ("albedo", capi.BUF_ALBEDO),
                           ("normal_depth", capi.BUF_NORMAL_DEPTH), ("indirect", capi.BUF_INDIRECT), ("combined", capi.BUF_COMBINED)):
            got = r.readback(kind)
            nbad = int((bits(got) != bits(ref[name])).any(-1).sum())
            assert nbad == 0, "%s (traversal %d, ggx %s): %d pixels differ" % (name, mode, ggx, nbad)
        s = r.stats()
        assert (s.rays_primary, s.rays_extension, s.rays_shadow) == ref["rays"] and s.guard_shade == 0 and s.guard_trace_any == 0
    # accumulation over frames and batches
    acc, rays = sc.render_accumulate(ocam, bluenoise, w, h, 0, 6, D, flags=O.FLAG_EXT_MATERIALS | O.FLAG_USE_BVH, threads=8)
    r.set_traversal(0)
    r.set_batch_paths(2 * w * h)
    r.accum_reset()
    r.stats_reset()
    r.render(0, 6, D, capi.RENDER_EXT_MATERIALS)
    assert np.array_equal(bits(r.readback(capi.BUF_ACCUM_SUM)[..., :3]), bits(acc[..., :3]))
    s = r.stats()  # AUTO = the small-scene kernels: the next-event rays are traced where they are generated, none is queued
    assert (s.rays_primary, s.rays_extension, s.rays_shadow) == rays and s.shadow_entries == 0 and s.launches_trace_any == 0
    r.close()


def test_ext_needs_materials(native_lib, bluenoise, cornell_path):
    r = capi.Renderer(0)
    r.upload_geometry(capi.Geometry(cornell_path))
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(32, 32)
    r.set_camera(capi.cornell_camera(32, 32))
    with pytest.raises(capi.CapError):
        r.render(0, 1, 2, capi.RENDER_EXT_MATERIALS)
    with pytest.raises(capi.CapError):
        r.upload_materials(np.zeros((3, 12), np.float32))  # one material per mesh
    r.close()


def test_ext_energy_sanity(native_lib, bluenoise, tmp_path):
    """Closed white furnace check of the estimator: a diffuse box (albedo 0.5) fully enclosing the camera, all six walls
    emitting L = 1: every pixel must converge to 1 + 0.5 + 0.25 + ... truncated at the depth = 2 - 0.5^(D+1)."""
    v = []
    f = []
    quads = [((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)), ((-1, -1, 1), (-1, 1, 1), (1, 1, 1), (1, -1, 1)),
             ((-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1)), ((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)),
             ((-1, -1, -1), (-1, -1, 1), (1, -1, 1), (1, -1, -1)), ((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1))]
    txt = ["o furnace"]
    for q in quads:
        n = np.cross(np.subtract(q[1], q[0]), np.subtract(q[3], q[0]))
        n = -n / np.linalg.norm(n) if np.dot(n, q[0]) > 0 else n / np.linalg.norm(n)  # inward normals
        base = len(v)
        for p in q:
            v.append(p)
            txt.append("v %g %g %g" % p)
        txt.append("vn %g %g %g" % tuple(n))
        f.append(base)
    for k, base in enumerate(f):
        txt.append("f %d//%d %d//%d %d//%d %d//%d" % (base + 1, k + 1, base + 2, k + 1, base + 3, k + 1, base + 4, k + 1))
    (tmp_path / "f.obj").write_text("\n".join(txt) + "\n")
    geo = capi.Geometry(str(tmp_path / "f.obj"))
    mats = np.zeros((1, 12), np.float32)
    mats[0, 0:3] = 0.5
    mats[0, 3] = 1.0
    mats[0, 8:11] = 1.0
    w, h, D, n = 32, 32, 6, 256
    r = capi.Renderer(0)
    r.upload_geometry(geo)
    r.upload_materials(mats)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    cam = capi.CameraData()
    cam.position[:] = (0.1, -0.2, 0.3)
    cam.forward[:] = (0, 0, -1)
    cam.right[:] = (-1, 0, 0)
    cam.up[:] = (0, 1, 0)
    cam.focal_length = 0.02
    cam.sensor_size[0] = 0.036
    cam.sensor_size[1] = 0.036
    r.set_resolution(w, h)
    r.set_camera(cam)
    r.render(0, n, D, capi.RENDER_EXT_MATERIALS)
    img = r.readback(capi.BUF_ACCUM_MEAN)[..., :3]
    expect = 2.0 - 0.5 ** (D + 1)
    assert abs(float(img.mean()) - expect) < 0.02, (float(img.mean()), expect)
    r.close()


def test_nee_pair_cull_is_exact_and_used(native_lib, bluenoise, tmp_path):
    """Next-event rays of the small-scene path skip the fan pairs that cannot occlude a segment between a scene point and the lamp
    (hull faces with the lamp at a safe distance inside; context.hip update_nee_pairs: rule + error bound against the contract's absolute
    tmin).  On the Cornell box that is the floor, the back and the two side walls -- NOT the ceiling, 1 cm above the lamp, whose rim
    rays graze it.  With the cull and without it (switch table, same context): the same bits in every plane, the accumulated image and
    the counters, over frames whose rays reach every corner; a lamp moved to the middle of the room keeps its pairs' count."""
    geo, mats = cornell_with_materials(tmp_path)
    w, h, D = 160, 120, 6
    r = capi.Renderer(0)
    r.upload_geometry(geo)
    r.upload_materials(mats)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(w, h)
    r.set_camera(capi.cornell_camera(w, h))
    r.set_traversal(2)
    v = r.debug_get(capi.Renderer.DEBUG_NEE_PAIRS)
    tested, pairs = v >> 32, v & 0xFFFFFFFF
    assert pairs == 16 and tested == 12, (tested, pairs)

    def frames():
        out = []
        for f in (0, 7, 1023):
            r.accum_reset()
            r.stats_reset()
            r.render(f, 1, D, capi.RENDER_AOV | capi.RENDER_EXT_MATERIALS)
            s = r.stats()
            assert s.guard_shade == 0 and s.guard_append == 0
            out.append([r.readback(k) for k in (capi.BUF_DIRECT, capi.BUF_INDIRECT, capi.BUF_COMBINED)] + [(s.rays_primary, s.rays_extension, s.rays_shadow)])
        r.accum_reset()
        r.render(0, 8, D, capi.RENDER_EXT_MATERIALS)
        out.append(r.readback(capi.BUF_ACCUM_SUM))
        return out

    with_cull = frames()
    r.debug_switch("CAP_NO_NEE_PAIR_CULL", 1)
    r.build_bvh()  # the list is made by cap_bvh_build / cap_materials_upload
    v = r.debug_get(capi.Renderer.DEBUG_NEE_PAIRS)
    assert v >> 32 == 16
    without = frames()
    for a, b in zip(with_cull[:3], without[:3]):
        assert a[3] == b[3]
        for x, y in zip(a[:3], b[:3]):
            assert np.array_equal(bits(x), bits(y))
    assert np.array_equal(bits(with_cull[3]), bits(without[3]))
    r.close()
