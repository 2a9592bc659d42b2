"""BASELINE.json configs 3 and 5 at their full image sizes, and the EXT shading model at depth 16.

configs[2]: cornell_box 3840x2160, depth 8, next-event estimation.  configs[4]: 4096x4096, depth 16, mixed Lambert / GGX /
emissive.  Both name shading features the reference does not have (GGX, emissive triangles, NEE): they run the EXT model
(DESIGN.md "EXT shading model"), whose specification is the oracle's shade_pixel_ext.  The sample counts of the configs (512 and
1024 spp) are bench-sized; here a few frames per size carry what does not depend on the count: every pixel of a crop of rows
bit-exact against the oracle, size-independent properties of the whole image (finite, sample count in .w, guards silent,
re-batching bit-identical, frame additivity) and the path-id / tile padding arithmetic at 2^24 pixels."""
import os
import shutil

import numpy as np
import pytest

from capsaicin_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ext_cornell(tmp_path_factory, bluenoise):
    from oracle import cap_oracle as O
    tmp = tmp_path_factory.mktemp("cornell_ext")
    txt = open(os.path.join(ROOT, "assets", "cornell_box.obj")).read().replace("mtllib cornellbox.mtl", "mtllib cornell_box.mtl")
    (tmp / "c.obj").write_text(txt)
    shutil.copy(os.path.join(ROOT, "assets", "cornell_box.mtl"), tmp / "cornell_box.mtl")
    geo = capi.Geometry(str(tmp / "c.obj"))
    mats = geo.materials()
    for m, (rough, ks) in capi.scene_config()["cornell_ggx"].items():  # "mixed Lambert/GGX/emissive"
        mats[int(m), 3] = rough
        mats[int(m), 4:7] = ks
    r = capi.Renderer(0)
    r.upload_geometry(geo)
    r.upload_materials(mats)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    sc = O.Scene(geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes, materials=mats)
    yield r, sc, O
    r.close()


def ocam(O, cam):
    return O.make_camera(tuple(cam.position), tuple(cam.forward), tuple(cam.right), tuple(cam.up), cam.sensor_size[0], cam.sensor_size[1],
                         cam.focal_length)


PLANES = (("gbuffer_geo", capi.BUF_GBUFFER_GEO), ("direct", capi.BUF_DIRECT), ("albedo", capi.BUF_ALBEDO),
          ("normal_depth", capi.BUF_NORMAL_DEPTH), ("indirect", capi.BUF_INDIRECT), ("combined", capi.BUF_COMBINED))


def test_ext_depth16_parity(ext_cornell, bluenoise):
    """EXT model at BASELINE configs[4]'s depth, every plane and the ray counters bit-exact, both traversals."""
    r, sc, O = ext_cornell
    w, h, D = 96, 80, 16
    cam = capi.cornell_camera(w, h)
    r.set_resolution(w, h)
    r.set_camera(cam)
    r.set_batch_paths(0)
    ref = sc.render_frame(ocam(O, cam), bluenoise, w, h, 3, D, flags=O.FLAG_EXT_MATERIALS)
    for mode in (1, 2):
        r.set_traversal(mode)
        r.accum_reset()
        r.stats_reset()
        r.render(3, 1, D, capi.RENDER_AOV | capi.RENDER_EXT_MATERIALS)
        for name, kind in PLANES:
            nbad = int((bits(r.readback(kind)) != bits(ref[name])).any(-1).sum())
            assert nbad == 0, "%s (traversal %d): %d pixels differ" % (name, mode, nbad)
        s = r.stats()
        assert (s.rays_primary, s.rays_extension, s.rays_shadow) == ref["rays"] and s.guard_shade == 0 and s.guard_trace_any == 0
    r.set_traversal(0)
    # reference shading model at depth 16 too (the loop bound is the only thing the depth changes there)
    from oracle import obj_oracle
    g = obj_oracle.load_geometry(os.path.join(ROOT, "assets", "cornell_box.obj"))
    sc0 = O.Scene(g["positions"], g["normals"], g["texcoords"], g["indices"], g["meshes"])
    ref0 = sc0.render_frame(ocam(O, cam), bluenoise, w, h, 3, D)
    r.accum_reset()
    r.render(3, 1, D, capi.RENDER_AOV)
    assert int((bits(r.readback(capi.BUF_COMBINED)) != bits(ref0["combined"])).any(-1).sum()) == 0


@pytest.mark.parametrize("w,h,D,spp", [(3840, 2160, 8, 3), (4096, 4096, 16, 2)], ids=["config3_2160p_nee", "config5_4096sq_depth16"])
def test_full_size(ext_cornell, bluenoise, w, h, D, spp):
    r, sc, O = ext_cornell
    flags = capi.RENDER_EXT_MATERIALS
    cam = capi.cornell_camera(w, h)
    r.set_traversal(0)
    r.set_resolution(w, h)
    r.set_camera(cam)
    r.set_batch_paths(0)
    r.accum_reset()
    r.stats_reset()
    r.render(0, spp, D, flags)
    a = r.readback(capi.BUF_ACCUM_SUM)
    s = r.stats()
    assert s.rays_primary == spp * w * h and s.rays_extension <= s.rays_primary * D and s.rays_shadow <= s.shaded_vertices
    assert s.guard_shade == 0 and s.guard_trace_any == 0
    assert np.isfinite(a).all() and (a[..., 3] == spp).all() and (a[..., :3] >= 0).all()
    assert float(a[..., :3].max()) > spp  # the lamp is in view: emission seen from the camera
    # re-batching: one frame per batch instead of all of them -> identical bits and counters
    r.set_batch_paths(w * h)
    r.accum_reset()
    r.stats_reset()
    r.render(0, spp, D, flags)
    s1 = r.stats()
    assert np.array_equal(bits(r.readback(capi.BUF_ACCUM_SUM)), bits(a))
    assert (s1.rays_extension, s1.rays_shadow, s1.shaded_vertices) == (s.rays_extension, s.rays_shadow, s.shaded_vertices)
    r.set_batch_paths(0)
    # additivity over frames in fp32 order
    tot = np.zeros_like(a)
    for f in range(spp):
        r.accum_reset()
        r.render(f, 1, D, flags)
        tot = tot + r.readback(capi.BUF_ACCUM_SUM)
    assert np.array_equal(bits(tot[..., :3]), bits(a[..., :3]))
    # oracle crops of the last frame: rows at the top edge, across the lamp / boxes, and the bottom edge (tile padding rows)
    r.accum_reset()
    r.stats_reset()
    r.render(spp - 1, 1, D, flags | capi.RENDER_AOV)
    got = {name: r.readback(kind) for name, kind in PLANES}
    for y0 in (0, (h // 3) & ~7, h - 8):
        ref = sc.render_frame(ocam(O, cam), bluenoise, w, h, spp - 1, D, flags=O.FLAG_EXT_MATERIALS | O.FLAG_USE_BVH, threads=16,
                              rows=(y0, y0 + 8))
        for name, _ in PLANES:
            nbad = int((bits(got[name][y0:y0 + 8]) != bits(ref[name][y0:y0 + 8])).any(-1).sum())
            assert nbad == 0, "%s rows %d..%d: %d pixels differ" % (name, y0, y0 + 7, nbad)


def test_reference_model_at_4096sq_depth16(native_lib, bluenoise, cornell_path):
    """The reference shading model at configs[4]'s size and depth: 2^24 pixels in the 26-bit local-pixel field of the path id."""
    from oracle import cap_oracle as O
    from oracle import obj_oracle
    w = h = 4096
    D = 16
    r = capi.Renderer(0)
    r.upload_geometry(capi.Geometry(cornell_path))
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    cam = capi.cornell_camera(w, h)
    r.set_resolution(w, h)
    r.set_camera(cam)
    r.render(0, 2, D, capi.RENDER_AOV)
    a = r.readback(capi.BUF_ACCUM_SUM)
    s = r.stats()
    assert np.isfinite(a).all() and (a[..., 3] == 2).all() and s.rays_primary == 2 * w * h and s.guard_shade == 0 and s.guard_trace_any == 0
    g = obj_oracle.load_geometry(cornell_path)
    sc = O.Scene(g["positions"], g["normals"], g["texcoords"], g["indices"], g["meshes"])
    got = r.readback(capi.BUF_COMBINED)
    for y0 in (0, 2048, h - 8):
        ref = sc.render_frame(ocam(O, cam), bluenoise, w, h, 1, D, flags=O.FLAG_USE_BVH, threads=16, rows=(y0, y0 + 8))
        assert int((bits(got[y0:y0 + 8]) != bits(ref["combined"][y0:y0 + 8])).any(-1).sum()) == 0
    r.close()
