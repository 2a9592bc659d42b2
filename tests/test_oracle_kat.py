"""Pins the CPU oracle: known answers of SURVEY.md 8c (hand-evaluated from the reference HLSL) and the reference's data
assets.  CPU only."""
import json
import math
import os

import numpy as np
import pytest

from oracle import cap_oracle as O
from oracle import obj_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))


def test_kd_default():
    assert abs(O.pow22(0.75) - KAT["kd_default"]) < 1e-6


def test_directional_light():
    for k in KAT["directional_light"]:
        d, i = O.directional_light(k["count"])
        np.testing.assert_allclose(d, k["dir"], atol=2e-7)
        np.testing.assert_allclose(i, k["intensity"], atol=2e-6)
    # period 4096 (lighting.h:22: count % 4096)
    a, b = O.directional_light(5), O.directional_light(5 + 4096)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_wang_hash():
    for x, y, h in KAT["wang_hash"]:
        assert O.wang_hash(x, y) == h


def test_bluenoise(bluenoise):
    assert [list(map(int, bluenoise[0, 0])), list(map(int, bluenoise[0, 1]))] == KAT["bluenoise_first_texels"]
    assert abs(float(bluenoise[..., 0].mean()) - 127.5) < 1e-9 and abs(float(bluenoise[..., 1].mean()) - 127.5) < 1e-9
    for k in KAT["bluenoise4x4"]:
        s = O.bluenoise4x4(bluenoise, k["xy"][0], k["xy"][1], k["count"])
        np.testing.assert_allclose(s, k["s"], atol=2e-7)
    # sampling.h:13-23: the sample stays in [0,1) and the 16 sub-texel offsets are visited over 16 counts
    seen = set()
    for c in range(16):
        s = O.bluenoise4x4(bluenoise, 3, 9, c)
        assert 0.0 <= s[0] < 1.0 and 0.0 <= s[1] < 1.0
        seen.add((c % 16) % 4 + 4 * ((c % 16) // 4))
    assert len(seen) == 16


def test_halton23():
    for i, p in enumerate(KAT["halton23"]):
        s = O.halton23(i)
        assert s[0] == np.float32(p[0]) and s[1] == np.float32(p[1])
        assert np.array_equal(O.halton23(i + 8), s)


def test_primary_rays_default_camera():
    k = KAT["default_camera_1080p"]
    sy = np.float32(0.036) * (np.float32(1080) / np.float32(1920))  # camera_system.cpp:10-17
    assert abs(float(sy) - k["sensor_y"]) < 1e-9
    cam = O.make_camera((0, 15, 0), (0, 0, 1), (1, 0, 0), (0, 1, 0), 0.036, float(sy), 0.016)
    for r in k["rays"]:
        o, d = O.primary_ray(cam, r["xy"][0], r["xy"][1], 1920, 1080, r["frame"])
        assert tuple(o) == (0.0, 15.0, 0.0)
        np.testing.assert_allclose(d, r["dir"], atol=3e-7)
        assert abs(float(np.linalg.norm(d.astype(np.float64))) - 1.0) < 2e-7


def test_sincos_contract_accuracy():
    xs = np.concatenate([np.linspace(0.0, 2 * math.pi, 4001), np.float32(2 * math.pi) * np.random.RandomState(1).rand(4000)])
    worst = 0.0
    for x in xs.astype(np.float32):
        s, c = O.sincos(float(x))
        worst = max(worst, abs(s - math.sin(float(x))), abs(c - math.cos(float(x))))
    assert worst < 2.5e-7, worst


def test_pow22_contract_accuracy():
    xs = np.concatenate([np.arange(1, 256) / 255.0, np.random.RandomState(2).rand(2000)]).astype(np.float32)
    for x in xs:
        ref = float(x) ** 2.2
        assert abs(O.pow22(float(x)) - ref) <= 4e-6 * max(ref, 1e-6), (x, O.pow22(float(x)), ref)
    assert O.pow22(0.0) == 0.0 and O.pow22(1.0) == 1.0 and O.pow22(-1.0) == 0.0


def test_map_to_hemisphere_properties(bluenoise):
    rs = np.random.RandomState(3)
    for _ in range(500):
        n = rs.randn(3)
        n /= np.linalg.norm(n)
        s = rs.rand(2).astype(np.float32)
        d = O.map_to_hemisphere(s, n.astype(np.float32)).astype(np.float64)
        assert abs(np.linalg.norm(d) - 1.0) < 3e-7
        cos_t = float(np.dot(d, n))
        # sampling.h:127: cos(theta) = sqrt(1 - r2)  (cosine-weighted hemisphere)
        assert abs(cos_t - math.sqrt(1.0 - float(s[1]))) < 3e-6
    # axis-aligned normals exercise both OrthoVector branches (sampling.h:95-108)
    for n in ((0, 0, 1), (0, 1, 0), (1, 0, 0), (0, -1, 0)):
        d = O.map_to_hemisphere((0.25, 0.5), n)
        assert np.all(np.isfinite(d)) and np.dot(d, n) > 0


def test_oct_encode():
    for n, e in (((0, 0, 1), (0.5, 0.5)), ((1, 0, 0), (1.0, 0.5)), ((0, 1, 0), (0.5, 1.0)), ((0, 0, -1), (1.0, 1.0))):
        np.testing.assert_allclose(O.oct_encode(n), e, atol=1e-7)


def test_triangle_intersection_rules():
    v0, v1, v2 = (0, 0, 0), (1, 0, 0), (0, 1, 0)
    t, u, v = O.intersect_triangle((0.25, 0.25, 1), (0, 0, -1), 0.0, 10.0, v0, v1, v2)
    assert abs(t - 1) < 2e-7 and abs(u - 0.25) < 1e-7 and abs(v - 0.25) < 1e-7  # barycentrics weight v1, v2
    # two-sided (tlas_system.cpp:47 TRIANGLE_CULL_DISABLE)
    assert O.intersect_triangle((0.25, 0.25, -1), (0, 0, 1), 0.0, 10.0, v0, v1, v2) is not None
    # tmin < t < tmax; t = T * rcp(det) with the contract's Newton-Raphson reciprocal (<= 1e-7 relative), so the interval ends
    # are sharp to about one ulp of t
    assert O.intersect_triangle((0.25, 0.25, 1), (0, 0, -1), 1.0001, 10.0, v0, v1, v2) is None
    assert O.intersect_triangle((0.25, 0.25, 1), (0, 0, -1), 0.0, 0.9999, v0, v1, v2) is None
    assert O.intersect_triangle((0.25, 0.25, 1), (0, 0, -1), 0.9999, 1.0001, v0, v1, v2) is not None
    # parallel ray and degenerate triangle never hit
    assert O.intersect_triangle((0.25, 0.25, 1), (1, 0, 0), 0.0, 10.0, v0, v1, v2) is None
    assert O.intersect_triangle((0.25, 0.25, 1), (0, 0, -1), 0.0, 10.0, v0, v0, v2) is None
    # outside
    assert O.intersect_triangle((0.75, 0.75, 1), (0, 0, -1), 0.0, 10.0, v0, v1, v2) is None


def test_cornell_fixture(cornell_path):
    g = obj_oracle.load_geometry(cornell_path)
    k = KAT["cornell"]
    assert g["meshes"].shape[0] == k["meshes"] and g["positions"].size // 3 == k["vertices"]
    assert g["indices"].size == k["indices"] and g["indices"].size // 3 == k["triangles"]
    p = g["positions"].reshape(-1, 3)
    np.testing.assert_allclose(p.min(0), k["bbox_lo"], atol=1e-6)
    np.testing.assert_allclose(p.max(0), k["bbox_hi"], atol=1e-6)
    # the mtllib name mismatch leaves every mesh untextured
    assert "cornellbox.mtl" in g["warn"] and np.all(g["meshes"][:, 5] == 0xFFFFFFFF)


def _random_scene(rs, ntri):
    c = rs.rand(ntri, 1, 3) * 4 - 2
    v = (c + (rs.rand(ntri, 3, 3) - 0.5) * 0.8).astype(np.float32)
    pos = v.reshape(-1, 3)
    nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    nrm = np.repeat(nrm[:, None, :], 3, axis=1).reshape(-1, 3).astype(np.float32)
    uv = rs.rand(ntri * 3, 2).astype(np.float32)
    idx = np.arange(ntri * 3, dtype=np.uint32)
    half = (ntri // 2) * 3
    meshes = np.array([[half, 0, half, 0, 0, 0xFFFFFFFF, 0, 0],
                       [ntri * 3 - half, half, ntri * 3 - half, half, 1, 0xFFFFFFFF, 0, 0]], np.uint32)
    idx[half:] -= half
    return pos, nrm, uv, idx, meshes


def test_bvh_equals_brute_force(bluenoise, cornell_path):
    g = obj_oracle.load_geometry(cornell_path)
    sc = O.Scene(g["positions"], g["normals"], g["texcoords"], g["indices"], g["meshes"])
    cam = O.make_camera((-0.01, 0.995, 3.4), (0, 0, -1), (-1, 0, 0), (0, 1, 0), 0.036, 0.036, 0.035)
    a = sc.render_frame(cam, bluenoise, 96, 96, 3, 4)
    b = sc.render_frame(cam, bluenoise, 96, 96, 3, 4, flags=O.FLAG_USE_BVH, threads=4)
    for k in ("gbuffer_geo", "direct", "albedo", "normal_depth", "indirect", "combined"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert a["rays"] == b["rays"]
    pos, nrm, uv, idx, meshes = _random_scene(np.random.RandomState(7), 300)
    sc = O.Scene(pos, nrm, uv, idx, meshes)
    cam = O.make_camera((0, 0, 6), (0, 0, -1), (-1, 0, 0), (0, 1, 0), 0.036, 0.036, 0.03)
    a = sc.render_frame(cam, bluenoise, 64, 64, 1, 3)
    b = sc.render_frame(cam, bluenoise, 64, 64, 1, 3, flags=O.FLAG_USE_BVH, threads=4)
    for k in ("gbuffer_geo", "indirect", "direct"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k


def test_render_semantics(bluenoise, cornell_path):
    g = obj_oracle.load_geometry(cornell_path)
    sc = O.Scene(g["positions"], g["normals"], g["texcoords"], g["indices"], g["meshes"])
    cam = O.make_camera((-0.01, 0.995, 3.4), (0, 0, -1), (-1, 0, 0), (0, 1, 0), 0.036, 0.036, 0.035)
    r = sc.render_frame(cam, bluenoise, 64, 64, 0, 2)
    geo = r["gbuffer_geo"].view(np.uint32)
    miss = geo[..., 2] == 0xFFFFFFFF
    assert miss.any() and (~miss).any()
    # rt_direct_lighting.hlsl:53-59 / rt_indirect.hlsl:75-79 for misses
    assert np.all(r["direct"][miss] == np.float32([0.7, 0.7, 0.85, 1.0])) and np.all(r["albedo"][miss] == 1.0)
    assert np.all(r["indirect"][miss] == np.float32([0, 0, 0, 1])) and np.all(r["normal_depth"][miss] == 0)
    assert np.all(geo[miss][:, 3] == 0xFFFFFFFF) and np.all(r["gbuffer_geo"][miss][:, :2] == 0)
    # hits: albedo is the untextured constant, instance ids are mesh slots, barycentrics inside the triangle
    kd = np.float32(O.pow22(0.75))
    assert np.all(r["albedo"][~miss][:, :3] == kd) and np.all(geo[~miss][:, 2] < 8)
    uv = r["gbuffer_geo"][~miss][:, :2]
    assert np.all(uv >= 0) and np.all(uv.sum(1) <= 1 + 1e-6)
    assert np.all(np.isfinite(r["combined"]))
    # combine_illumination.hlsl:29
    np.testing.assert_array_equal(r["combined"][..., :3], r["indirect"][..., :3] * r["albedo"][..., :3] + r["direct"][..., :3])
    # num_bounces = 0: one indirect sample of sky/none, no extension rays beyond ... (rt_indirect.hlsl:91)
    r0 = sc.render_frame(cam, bluenoise, 64, 64, 0, 0)
    assert r0["rays"][1] == 0 and np.all(r0["indirect"][..., :3] == 0)
    # accumulate = running sum of combined in frame order
    acc, rays = sc.render_accumulate(cam, bluenoise, 32, 32, 5, 3, 1)
    s = np.zeros((32, 32, 4), np.float32)
    for f in range(5, 8):
        s = s + sc.render_frame(cam, bluenoise, 32, 32, f, 1)["combined"]
    assert np.array_equal(acc, s)


def test_degenerate_normals_give_black_not_nan(bluenoise):
    # an OBJ without vn gets all-zero normals (asset_load_system.cpp:124-129): normalize(0) is NaN, HLSL max() drops it
    pos = np.float32([[-1, -1, 0], [1, -1, 0], [0, 1, 0]])
    sc = O.Scene(pos, np.zeros_like(pos), np.zeros((3, 2), np.float32), np.uint32([0, 1, 2]),
                 np.uint32([[3, 0, 3, 0, 0, 0xFFFFFFFF, 0, 0]]))
    cam = O.make_camera((0, 0, 3), (0, 0, -1), (-1, 0, 0), (0, 1, 0), 0.036, 0.036, 0.05)
    r = sc.render_frame(cam, bluenoise, 16, 16, 0, 2)
    hit = r["gbuffer_geo"].view(np.uint32)[..., 2] == 0
    assert hit.any()
    assert np.all(r["direct"][hit][:, :3] == 0) and np.all(r["indirect"][hit][:, :3] == 0)


def test_texture_sampling():
    tex = np.zeros((2, 2, 4), np.uint8)
    tex[0, 0] = (255, 0, 0, 255)
    tex[0, 1] = (0, 255, 0, 255)
    tex[1, 0] = (0, 0, 255, 255)
    tex[1, 1] = (255, 255, 255, 255)
    # texel centres reproduce the texel; WRAP addressing (d3dx12.h:943-944) blends across the border
    np.testing.assert_allclose(O.sample_texture(tex, 0.25, 0.25), (1, 0, 0), atol=1e-7)
    np.testing.assert_allclose(O.sample_texture(tex, 0.75, 0.25), (0, 1, 0), atol=1e-7)
    np.testing.assert_allclose(O.sample_texture(tex, 0.5, 0.25), (0.5, 0.5, 0), atol=1e-7)
    np.testing.assert_allclose(O.sample_texture(tex, 0.0, 0.25), (0.5, 0.5, 0), atol=1e-7)
    np.testing.assert_allclose(O.sample_texture(tex, 1.25, -0.75), O.sample_texture(tex, 0.25, 0.25), atol=1e-6)


def test_unorm8_is_the_division():
    """kernels.hip sample_texture() turns a texel byte into b / 255.0f with a multiply and one residual step (two fmaf) instead of
    the division the oracle performs (cap_oracle.cpp sample_texture): equal for all 256 bytes, in exact arithmetic rounded to fp32
    once per operation (round to nearest even), which is what v_mul_f32 / v_fma_f32 and fmaf do."""
    from fractions import Fraction

    def rn32(x):  # correctly rounded fp32 value of a rational, as a Fraction
        if x == 0:
            return Fraction(0)
        s, x = (-1 if x < 0 else 1), abs(x)
        e = 0
        while x >= 2:
            x, e = x / 2, e + 1
        while x < 1:
            x, e = x * 2, e - 1
        m = x * (1 << 23)           # 1.xxx * 2^23: 24-bit significand plus a fraction
        n, frac = int(m), m - int(m)
        if frac > Fraction(1, 2) or (frac == Fraction(1, 2) and n & 1):
            n += 1
        return s * Fraction(n, 1 << 23) * Fraction(2) ** e

    r = rn32(Fraction(1, 255))
    assert float(r) == float(np.float32(1.0) / np.float32(255.0))
    for b in range(256):
        fb = Fraction(b)
        q = rn32(fb * r)
        e = rn32(-q * 255 + fb)       # fmaf(-q, 255, b)
        got = rn32(e * r + q)         # fmaf(e, r, q)
        assert got == rn32(fb / 255), b
        assert float(got) == float(np.float32(b) / np.float32(255.0))
