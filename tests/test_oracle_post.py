"""CPU checks of the oracle's reconstruction chain (oracle/cap_oracle_post.cpp).  The reference ships no test
or golden image for these passes (parity unpinned, see the oracle's header), so the chain is pinned by properties that follow
from the reference's shader text: pass-through of sky pixels, fixed point on constant input, variance reduction, determinism."""
import numpy as np
import pytest

from oracle import cap_oracle as O


def camera(w, h):
    return O.make_camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (-1.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.036, 0.036 * h / w, 0.035)


def wall_planes(w, h, cam, rng=None, value=0.5):
    """A wall facing the camera at z = -2: normal_depth = (oct(+z), instance 0, |cam - p|)."""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    cx = ((xs + 0.5) / w - 0.5) * cam.sensor_size[0]
    cy = ((ys + 0.5) / h - 0.5) * cam.sensor_size[1]
    d = np.stack([-cx, cy, -np.full_like(cx, cam.focal_length)], -1)  # right = -x, up = +y, forward = -z
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    depth = (2.0 / -d[..., 2]).astype(np.float32)
    nd = np.zeros((h, w, 4), np.float32)
    nd[..., 0:2] = 0.5  # oct_encode((0, 0, 1)) = (0.5, 0.5)
    nd[..., 3] = depth
    ind = np.full((h, w, 4), value, np.float32)
    if rng is not None:
        ind[..., :3] = rng.uniform(0.0, 1.0, (h, w, 1)).astype(np.float32)
    alb = np.ones((h, w, 4), np.float32)
    alb[..., 3] = 0.0
    dr = np.zeros((h, w, 4), np.float32)
    return {"indirect": ind, "direct": dr, "albedo": alb, "normal_depth": nd}


def test_sky_pixels_pass_through():
    """depth < 1e-5 short-circuits every pass (spatial_gather.hlsl:53, temporal_accumulation.hlsl:232, eaw_blur.hlsl:66/160,
    temporal_accumulation.hlsl:378): the output is min(indirect, 10) * albedo + direct, exactly.  The last row and column are
    left out: UVtoXY clamps to dim - 1 (utils.h:6-10), so SampleBilinear at their pixel centres blends two texels."""
    w, h = 40, 24
    rng = np.random.default_rng(1)
    cam = camera(w, h)
    planes = {k: rng.uniform(0.0, 12.0, (h, w, 4)).astype(np.float32) for k in ("indirect", "direct", "albedo")}
    planes["normal_depth"] = np.zeros((h, w, 4), np.float32)
    chain = O.PostChain(w, h)
    for f in range(3):
        out = chain.frame(O.PostSettings(), f, cam, cam, planes)
        want = np.minimum(planes["indirect"][..., :3], np.float32(10.0)) * planes["albedo"][..., :3] + planes["direct"][..., :3]
        assert np.array_equal(out[:-1, :-1, :3].view(np.uint32), want.astype(np.float32)[:-1, :-1].view(np.uint32))
        assert not np.array_equal(out[-1, :, :3], want[-1])  # the clamp quirk is reproduced, not fixed
        assert np.all(out[..., 3] == 1.0)


@pytest.mark.parametrize("settings", [dict(), dict(gather=0), dict(denoise=0), dict(eaw5=0)])
def test_constant_input_is_a_fixed_point(settings):
    w, h = 48, 32
    cam = camera(w, h)
    planes = wall_planes(w, h, cam, value=0.37)
    chain = O.PostChain(w, h)
    for f in range(5):
        out = chain.frame(O.PostSettings(**settings), f, cam, cam, planes)
        assert np.all(np.isfinite(out))
        # not tighter: TAA's clip box is mean +- 5 * sqrt(|E[v^2] - E[v]^2|), and on a constant image that difference is rounding
        # noise of order 1e-8, i.e. a box of ~5e-4 around the colour that the zero history of frame 0 is clipped to
        assert np.abs(out[..., :3] - 0.37).max() < 5e-3, f
    assert np.abs(out[..., :3] - 0.37).max() < 1.5e-3  # and the residual decays with the TAA feedback (x0.9 per frame)


def test_denoiser_reduces_variance_and_keeps_the_mean():
    w, h = 64, 48
    cam = camera(w, h)
    rng = np.random.default_rng(7)
    chain = O.PostChain(w, h)
    for f in range(8):
        planes = wall_planes(w, h, cam, rng=rng)
        out = chain.frame(O.PostSettings(), f, cam, cam, planes)
    inner = out[8:-8, 8:-8, 0]
    assert inner.std() < 0.25 * planes["indirect"][..., 0].std()
    assert abs(float(inner.mean()) - 0.5) < 0.05


def test_history_length_and_reset_on_camera_cut():
    """A camera jump that moves the wall out of the previous view resets the history (temporal_accumulation.hlsl:247-259):
    the frame after the cut equals a first frame."""
    w, h = 32, 24
    cam = camera(w, h)
    far = O.make_camera((100.0, 0.0, 0.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.036, 0.036 * h / w, 0.035)
    rng = np.random.default_rng(3)
    seq = [wall_planes(w, h, cam, rng=rng) for _ in range(4)]
    s = O.PostSettings(denoise=0, gather=0)
    a = O.PostChain(w, h)
    for f in range(3):
        a.frame(s, f, cam, cam, seq[f])
    cut = a.frame(s, 3, cam, far, seq[3])  # previous camera looks away: every reprojection leaves the image
    # no history is usable: accumulate resets, TAA writes the bilinear tap of the combined image = the image itself
    want = seq[3]["indirect"][..., :3] * seq[3]["albedo"][..., :3] + seq[3]["direct"][..., :3]
    assert np.array_equal(cut[:-1, :-1, :3], want[:-1, :-1])


def test_lowres_indirect_is_the_interleaved_quarter(bluenoise, cornell_path, native_lib):
    """LOWRES_INDIRECT (rt_indirect.hlsl:53-59): the half-resolution indirect image of frame f holds exactly the full-resolution
    pass's pixels at sp_offset = ((f % 4) / 2, (f % 4) % 2); a quarter of the extension rays are traced; direct lighting and
    the G-buffer stay full resolution.  Four consecutive frames cover every pixel once."""
    from capsaicin_amd import capi
    geo = capi.Geometry(cornell_path)
    sc = O.Scene(geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes)
    w, h, D = 64, 48, 2
    cam = capi.cornell_camera(w, h)
    oc = O.make_camera(tuple(cam.position), tuple(cam.forward), tuple(cam.right), tuple(cam.up), cam.sensor_size[0], cam.sensor_size[1],
                       cam.focal_length)
    seen = np.zeros((h, w), bool)
    for f in range(4):
        low = sc.render_frame(oc, bluenoise, w, h, f, D, flags=O.FLAG_LOWRES_INDIRECT, threads=4)
        full = sc.render_frame(oc, bluenoise, w, h, f, D, threads=4)
        ox, oy = (f % 4) // 2, (f % 4) % 2
        assert np.array_equal(low["indirect_lowres"], full["indirect"][oy::2, ox::2])
        for k in ("direct", "albedo", "normal_depth", "gbuffer_geo"):
            assert np.array_equal(low[k].view(np.uint32), full[k].view(np.uint32))  # gbuffer_geo carries ids as float bits
        assert low["rays"][0] == full["rays"][0] and low["rays"][1] < 0.3 * full["rays"][1]
        seen[oy::2, ox::2] = True
    assert seen.all()
    with pytest.raises(RuntimeError):  # odd extents have no 2x2 blocks
        sc.render_frame(oc, bluenoise, w + 1, h, 0, D, flags=O.FLAG_LOWRES_INDIRECT)


def test_lowres_chain_keeps_unsampled_history():
    """UPSCALE2X Accumulate (temporal_accumulation.hlsl:307-313): a pixel without a new sample this frame keeps its colour
    history (alpha = 1) and its history length; the sampled pixel of each 2x2 block blends as usual."""
    w, h = 32, 24
    cam = camera(w, h)
    s = O.PostSettings(lowres_indirect=1, gather=0, denoise=0)
    chain = O.PostChain(w, h)
    rng = np.random.default_rng(5)
    outs = []
    for f in range(6):
        planes = wall_planes(w, h, cam)
        planes["indirect_lowres"] = rng.uniform(0.2, 0.8, (h // 2, w // 2, 4)).astype(np.float32)
        outs.append(chain.frame(s, f, cam, cam, planes))
        assert np.all(np.isfinite(outs[-1]))
    # constant lowres input is a fixed point of the upscaling chain as well
    chain = O.PostChain(w, h)
    for f in range(5):
        planes = wall_planes(w, h, cam)
        planes["indirect_lowres"] = np.full((h // 2, w // 2, 4), 0.37, np.float32)
        out = chain.frame(O.PostSettings(lowres_indirect=1), f, cam, cam, planes)
    assert np.abs(out[..., :3] - 0.37).max() < 1.5e-3


def test_deterministic():
    w, h = 32, 32
    cam = camera(w, h)
    outs = []
    for _ in range(2):
        rng = np.random.default_rng(11)
        chain = O.PostChain(w, h)
        for f in range(3):
            out = chain.frame(O.PostSettings(), f, cam, cam, wall_planes(w, h, cam, rng=rng))
        outs.append(out)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


def test_output_selector_on_sky_pixels():
    """SettingsComponent::output = CombineIllumination's `type` (combine_illumination.hlsl:26-40; raytracing_system.cpp:1415).  On
    sky pixels every other pass is a pass-through (see test_sky_pixels_pass_through), so the chain's output IS Combine's:
    0 indirect * albedo + direct, 1 direct, 2 (indirect.xyz, 1), 3 indirect.www -- and a sky pixel's variance channel is what
    Accumulate left there, 0 (temporal_accumulation.hlsl:232 returns before the moments are touched)."""
    w, h = 40, 24
    rng = np.random.default_rng(7)
    cam = camera(w, h)
    planes = {k: rng.uniform(0.0, 8.0, (h, w, 4)).astype(np.float32) for k in ("indirect", "direct", "albedo")}
    planes["normal_depth"] = np.zeros((h, w, 4), np.float32)
    ind = np.minimum(planes["indirect"][..., :3], np.float32(10.0))
    want = {0: ind * planes["albedo"][..., :3] + planes["direct"][..., :3], 1: planes["direct"][..., :3], 2: ind}
    for output in (0, 1, 2, 3):
        chain = O.PostChain(w, h)
        out = chain.frame(O.PostSettings(output=output), 0, cam, cam, planes)
        if output == 3:
            assert np.all(out[:-1, :-1, 0] == out[:-1, :-1, 1]) and np.all(out[:-1, :-1, 1] == out[:-1, :-1, 2])  # .www
        else:
            assert np.array_equal(out[:-1, :-1, :3].view(np.uint32), want[output].astype(np.float32)[:-1, :-1].view(np.uint32)), output
    with pytest.raises(RuntimeError):
        O.PostChain(w, h).frame(O.PostSettings(output=4), 0, cam, cam, planes)
