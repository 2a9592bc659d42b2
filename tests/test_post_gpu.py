"""Reconstruction chain (Gather -> Accumulate -> BlurDisocclusion -> Blur -> Combine -> TAA) on the GPU against
the oracle's chain (oracle/cap_oracle_post.cpp), bit for bit, frame after frame with the histories carried along."""
import numpy as np
import pytest

from capsaicin_amd import capi

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def ocam_of(O, cam):
    return O.make_camera(tuple(cam.position), tuple(cam.forward), tuple(cam.right), tuple(cam.up), cam.sensor_size[0], cam.sensor_size[1],
                         cam.focal_length)


def moved(cam, dx, dy, dz):
    c = capi.CameraData.from_buffer_copy(bytes(cam))
    c.position[0] += dx
    c.position[1] += dy
    c.position[2] += dz
    return c


@pytest.mark.parametrize("settings", [dict(), dict(gather=0, eaw5=0), dict(denoise=0), dict(eaw_luma_sigma=1.5, gather_normal_sigma=16.0),
                                      dict(use_variance=0), dict(use_variance=0, eaw5=0),  # RaytracingOptions::use_variance, raytracing_system.h:25
                                      # SettingsComponent::output -> CombineIllumination's `type` (combine_illumination.hlsl:26-40,
                                      # raytracing_system.cpp:1415): direct, indirect, variance (.www after the last blur / without one)
                                      dict(output=1), dict(output=2), dict(output=3), dict(output=3, denoise=0), dict(output=2, eaw5=0),
                                      # sigmas outside the range post.hip's unscaled division is used on: every tile takes the IEEE form
                                      # (the fallback of docs/experiments.md (68)); a depth sigma of 0 is the reference's "weight 1"
                                      dict(eaw_depth_sigma=1e-13, gather_depth_sigma=1e-13), dict(eaw_luma_sigma=1e13, gather_luma_sigma=1e-14),
                                      dict(eaw_depth_sigma=0.0, gather_depth_sigma=0.0)])
def test_post_chain_parity_cornell(native_lib, bluenoise, cornell_path, settings):
    from oracle import cap_oracle as O
    w, h, D = 150, 101, 2  # not multiples of the 32x8 workgroup footprint or the 8x8 render tiles
    geo = capi.Geometry(cornell_path)
    r = capi.Renderer(0)
    r.upload_geometry(geo)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(w, h)
    sc = O.Scene(geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes)
    chain = O.PostChain(w, h)
    base = capi.cornell_camera(w, h)
    # static for four frames, then a dolly/strafe (reprojection, disocclusion at the box edges, the non-static TAA branch)
    cams = [base] * 4 + [moved(base, 0.02 * k, 0.01 * k, -0.03 * k) for k in range(1, 4)] + [moved(base, 0.06, 0.03, -0.09)] * 2
    gs, os_ = capi.PostSettings(**settings), O.PostSettings(**settings)
    prev = cams[0]
    for f, cam in enumerate(cams):
        r.set_camera(cam)
        r.render(f, 1, D, capi.RENDER_AOV)
        r.post_frame(gs, f, prev)
        got = r.post_readback()
        ref = sc.render_frame(ocam_of(O, cam), bluenoise, w, h, f, D, threads=8)
        want = chain.frame(os_, f, ocam_of(O, cam), ocam_of(O, prev), ref)
        assert np.all(np.isfinite(got))
        nbad = int((bits(got) != bits(want)).any(-1).sum())
        assert nbad == 0, "frame %d (%s): %d pixels differ, max abs %g" % (f, settings, nbad, float(np.abs(got - want).max()))
        prev = cam
    # the denoised image is smoother than the raw one-sample frame but keeps its mean
    raw = ref["combined"][..., :3]
    if settings.get("output", 0) == 0:  # (the other output types show one term of the sum, or its variance)
        assert abs(float(got[..., :3].mean()) - float(raw.mean())) < 0.1 * float(raw.mean()) + 0.02
    r.close()


def test_unscaled_division_is_the_ieee_division(native_lib):
    """The exact mode's per-tap divisions skip the scaling steps of hipcc's IEEE expansion where the operands cannot trigger them
    (post.hip div_unscaled; the stencil kernels establish the range per tile while staging and take the IEEE form otherwise).  Both forms
    on the device, bit for bit: every positive normal float through the contract's log2, 2^30 operand pairs over the whole stated range."""
    r = capi.Renderer(0)
    assert r.debug_get(capi.Renderer.DEBUG_SELFTEST_DIV) == 0
    r.close()


@pytest.mark.parametrize("settings", [dict(), dict(use_variance=0), dict(eaw5=0, gather_luma_sigma=1.0)])
def test_fast_weights_within_stated_tolerance(native_lib, bluenoise, cornell_path, settings):
    """CapPostSettings::fast_weights (hardware exp2 / log2 / rcp in the edge-stopping weights; not a reference option) against the
    ORACLE's exact chain fed with the same ray-pass frames, over a nine-frame sequence with camera motion; the fast chain carries its
    own histories, so the bounds cover the drift the temporal feedback accumulates.  The stated tolerance (include/capsaicin_hip.h),
    per colour channel with e = |fast - exact| / (|exact| + 1e-3):
        median e <= 2e-5,   99 % of the channels e <= 4e-3,   every channel e <= 3e-2.
    Why a distribution and not one number: the weights themselves agree to ~1e-5 (the exact mode's polynomials are the LESS accurate
    side), but TAA clips the history to mean +- scale * sqrt(|m2 - m1^2|) of a 5x5 neighbourhood (temporal_accumulation.hlsl:98-137):
    in a flat region the variance is a cancellation residue, its square root turns a 1e-5 input difference into a 1e-2 box
    difference, and the static branch feeds 98 % of it back.  Measured (tools/fast_err.py; round 5, with the fast IntegrateTemporally and
    TAA values): median 8e-6, 99th percentile 2.4e-3, maximum 1.6e-2.  The reference's own RGBA16F storage perturbs the same inputs by 5e-4.  The exact mode is held to 0 above."""
    from oracle import cap_oracle as O
    w, h, D = 150, 101, 2
    geo = capi.Geometry(cornell_path)
    r = capi.Renderer(0)
    r.upload_geometry(geo)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(w, h)
    sc = O.Scene(geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes)
    chain = O.PostChain(w, h)
    base = capi.cornell_camera(w, h)
    cams = [base] * 4 + [moved(base, 0.02 * k, 0.01 * k, -0.03 * k) for k in range(1, 4)] + [moved(base, 0.06, 0.03, -0.09)] * 2
    gs, os_ = capi.PostSettings(fast_weights=1, **settings), O.PostSettings(**settings)
    prev, differs = cams[0], False
    for f, cam in enumerate(cams):
        r.set_camera(cam)
        r.render(f, 1, D, capi.RENDER_AOV)
        r.post_frame(gs, f, prev)
        got = r.post_readback()
        ref = sc.render_frame(ocam_of(O, cam), bluenoise, w, h, f, D, threads=8)
        want = chain.frame(os_, f, ocam_of(O, cam), ocam_of(O, prev), ref)
        assert np.all(np.isfinite(got))
        e = np.abs(got.astype(np.float64) - want)[..., :3] / (np.abs(want[..., :3]) + 1e-3)
        assert np.median(e) <= 2e-5 and np.percentile(e, 99) <= 4e-3 and e.max() <= 3e-2, \
            "frame %d (%s): median %.2e p99 %.2e max %.2e" % (f, settings, np.median(e), np.percentile(e, 99), e.max())
        differs |= bool((bits(got) != bits(want)).any())
        prev = cam
    assert differs  # it IS another arithmetic: equal bits would mean the switch does nothing
    r.close()


@pytest.mark.parametrize("traversal", [1, 2])
def test_gbuffer_feedback_parity(native_lib, bluenoise, cornell_path, traversal):
    """The reference's default configuration (RaytracingOptions::gbuffer_feedback = true, raytracing_system.h:26): the indirect
    pass reads the previous frame's reconstruction output (rt_indirect.hlsl:116-145), so render and chain feed each other frame
    after frame.  Ray passes and chain output are compared bit for bit with the oracle running the same loop."""
    from oracle import cap_oracle as O
    w, h, D = 120, 90, 3
    geo = capi.Geometry(cornell_path)
    r = capi.Renderer(0)
    r.upload_geometry(geo)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_traversal(traversal)  # 1: LBVH kernels + stand-alone shade, 2: fused exhaustive kernels
    r.set_resolution(w, h)
    sc = O.Scene(geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes)
    chain = O.PostChain(w, h)
    base = capi.cornell_camera(w, h)
    cams = [base] * 3 + [moved(base, 0.03 * k, 0.0, -0.05 * k) for k in range(1, 3)] + [moved(base, 0.06, 0.0, -0.1)]
    gs, os_ = capi.PostSettings(), O.PostSettings()
    prev = cams[0]
    prev_nd = np.zeros((h, w, 4), np.float32)
    hist = np.zeros((h, w, 4), np.float32)
    reused_any = False
    for f, cam in enumerate(cams):
        r.set_camera(cam)
        r.set_prev_camera(prev)
        r.stats_reset()
        r.render(f, 1, D, capi.RENDER_AOV | capi.RENDER_GBUFFER_FEEDBACK)
        ref = sc.render_frame(ocam_of(O, cam), bluenoise, w, h, f, D, threads=8, feedback=(ocam_of(O, prev), prev_nd, hist))
        plain = sc.render_frame(ocam_of(O, cam), bluenoise, w, h, f, D, threads=8)
        for name, kind in (("direct", capi.BUF_DIRECT), ("albedo", capi.BUF_ALBEDO), ("normal_depth", capi.BUF_NORMAL_DEPTH),
                           ("indirect", capi.BUF_INDIRECT)):
            got = r.readback(kind)
            nbad = int((bits(got) != bits(ref[name])).any(-1).sum())
            assert nbad == 0, "frame %d %s: %d pixels differ" % (f, name, nbad)
        s = r.stats()
        assert (s.rays_primary, s.rays_extension, s.rays_shadow) == ref["rays"]
        if f == 0:  # cleared histories: every vertex is a disocclusion, the frame equals the plain path tracer's
            assert ref["rays"] == plain["rays"] and np.array_equal(bits(ref["indirect"]), bits(plain["indirect"]))
        else:
            reused_any |= ref["rays"][1] < plain["rays"][1]  # paths end early at vertices the last frame saw
        r.post_frame(gs, f, prev)
        got = r.post_readback()
        want = chain.frame(os_, f, ocam_of(O, cam), ocam_of(O, prev), ref)
        nbad = int((bits(got) != bits(want)).any(-1).sum())
        assert nbad == 0, "frame %d chain output: %d pixels differ" % (f, nbad)
        prev, prev_nd, hist = cam, ref["normal_depth"], want
    assert reused_any
    # misuse
    with pytest.raises(capi.CapError, match="one frame per call"):
        r.render(0, 2, D, capi.RENDER_GBUFFER_FEEDBACK)
    r.close()


@pytest.mark.parametrize("feedback", [False, True])
def test_lowres_indirect_parity(native_lib, bluenoise, cornell_path, feedback):
    """RaytracingOptions::lowres_indirect: the indirect pass on the half-resolution grid, 2x2-interleaved over
    four frames (rt_indirect.hlsl:53-59), Gather and Accumulate in their UPSCALE2X form.  Ray passes, the half-resolution image
    and the chain output against the oracle, bit for bit, over two interleave cycles with camera motion."""
    from oracle import cap_oracle as O
    w, h, D = 136, 90, 2
    geo = capi.Geometry(cornell_path)
    r = capi.Renderer(0)
    r.upload_geometry(geo)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(w, h)
    sc = O.Scene(geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes)
    chain = O.PostChain(w, h)
    base = capi.cornell_camera(w, h)
    cams = [base] * 5 + [moved(base, 0.02 * k, 0.01 * k, -0.03 * k) for k in range(1, 5)]
    gs, os_ = capi.PostSettings(lowres_indirect=1), O.PostSettings(lowres_indirect=1)
    prev, prev_nd, hist = cams[0], np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    flags = capi.RENDER_AOV | capi.RENDER_LOWRES_INDIRECT | (capi.RENDER_GBUFFER_FEEDBACK if feedback else 0)
    for f, cam in enumerate(cams):
        r.set_camera(cam)
        r.set_prev_camera(prev)
        r.stats_reset()
        r.render(f, 1, D, flags)
        ref = sc.render_frame(ocam_of(O, cam), bluenoise, w, h, f, D, flags=O.FLAG_LOWRES_INDIRECT, threads=8,
                              feedback=(ocam_of(O, prev), prev_nd, hist) if feedback else None)
        for name, kind in (("direct", capi.BUF_DIRECT), ("albedo", capi.BUF_ALBEDO), ("normal_depth", capi.BUF_NORMAL_DEPTH),
                           ("indirect_lowres", capi.BUF_INDIRECT_LOWRES)):
            got = r.readback(kind)
            assert got.shape == ref[name].shape
            nbad = int((bits(got) != bits(ref[name])).any(-1).sum())
            assert nbad == 0, "frame %d %s: %d pixels differ" % (f, name, nbad)
        s = r.stats()
        assert (s.rays_primary, s.rays_extension, s.rays_shadow) == ref["rays"]
        assert s.rays_extension < 0.5 * w * h  # a quarter of the pixels start an indirect path
        r.post_frame(gs, f, prev)
        got = r.post_readback()
        want = chain.frame(os_, f, ocam_of(O, cam), ocam_of(O, prev), ref)
        nbad = int((bits(got) != bits(want)).any(-1).sum())
        assert nbad == 0, "frame %d chain output: %d pixels differ, max abs %g" % (f, nbad, float(np.abs(got - want).max()))
        prev, prev_nd, hist = cam, ref["normal_depth"], want
    # misuse
    with pytest.raises(capi.CapError, match="lowres_indirect"):
        r.post_frame(capi.PostSettings(), len(cams) - 1, prev)
    r.set_resolution(w + 1, h)
    with pytest.raises(capi.CapError, match="even"):
        r.render(0, 1, D, capi.RENDER_LOWRES_INDIRECT)
    r.close()


def test_post_chain_reset_and_errors(native_lib, bluenoise, cornell_path):
    w, h = 64, 48
    r = capi.Renderer(0)
    r.upload_geometry(capi.Geometry(cornell_path))
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(w, h)
    cam = capi.cornell_camera(w, h)
    r.set_camera(cam)
    s = capi.PostSettings()
    with pytest.raises(capi.CapError, match="CAP_RENDER_AOV"):
        r.post_frame(s, 0, cam)
    with pytest.raises(capi.CapError, match="has not run"):
        r.post_readback()
    r.render(0, 1, 1, capi.RENDER_AOV)
    with pytest.raises(capi.CapError, match="luma"):
        r.post_frame(capi.PostSettings(eaw_luma_sigma=0.0), 0, cam)
    r.post_frame(s, 0, cam)
    first = r.post_readback()
    r.render(1, 1, 1, capi.RENDER_AOV)
    r.post_frame(s, 1, cam)
    second = r.post_readback()
    assert not np.array_equal(first, second)
    # a reset sequence reproduces the first frame exactly
    r.post_reset()
    r.render(0, 1, 1, capi.RENDER_AOV)
    r.post_frame(s, 0, cam)
    assert np.array_equal(bits(r.post_readback()), bits(first))
    r.set_shard(0, 2)
    r.render(0, 1, 1, capi.RENDER_AOV)
    with pytest.raises(capi.CapError, match="unsharded"):
        r.post_frame(s, 0, cam)
    r.close()


@pytest.mark.parametrize("count,lowres", [(2, False), (3, False), (8, False), (3, True)])
def test_post_chain_on_gathered_shards(native_lib, bluenoise, cornell_path, count, lowres):
    """on sharded contexts the chain runs on one rank on the gathered ray-pass outputs.  Every shard renders its
    tiles, packs the four chain inputs (cap_resolve_aov_tiles), the buffers are concatenated rank-major (what one gather delivers)
    and the root runs cap_post_frame_gathered: bit-identical to the unsharded render + cap_post_frame, frame after frame."""
    import torch
    w, h, D = (152, 100, 2) if lowres else (150, 101, 2)  # the half-resolution indirect pass needs even sizes
    geo = capi.Geometry(cornell_path)
    r = capi.Renderer(0)
    r.upload_geometry(geo)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(w, h)
    base = capi.cornell_camera(w, h)
    cams = [base] * 2 + [moved(base, 0.02 * k, 0.01 * k, -0.03 * k) for k in range(1, 4)]
    s = capi.PostSettings(lowres_indirect=1) if lowres else capi.PostSettings()
    aov = capi.RENDER_AOV | (capi.RENDER_LOWRES_INDIRECT if lowres else 0)

    def sequence(sharded):
        out, prev = [], cams[0]
        r.post_reset()
        for f, cam in enumerate(cams):
            r.set_camera(cam)
            if not sharded:
                r.set_shard(0, 1)
                r.render(f, 1, D, aov)
                r.post_frame(s, f, prev)
            else:
                bufs = []
                for idx in range(count):
                    r.set_shard(idx, count)
                    r.render(f, 1, D, aov)
                    n = r.aov_tile_buffer_floats()
                    assert n == 4 * r.tile_buffer_floats()
                    t = torch.empty(n, dtype=torch.float32, device="cuda")
                    torch.cuda.synchronize()  # the renderer runs on its own stream
                    r.resolve_aov_tiles(t.data_ptr())
                    r.sync()
                    bufs.append(t)
                gathered = torch.cat(bufs)
                torch.cuda.synchronize()
                r.post_frame_gathered(s, f, prev, gathered.data_ptr(), count)
            out.append(r.post_readback())
            prev = cam
        return out

    want = sequence(False)
    got = sequence(True)
    for f, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(bits(a), bits(b)), "frame %d: %d pixels differ" % (f, int((bits(a) != bits(b)).any(-1).sum()))
    # error paths: a plain cap_post_frame refuses the sharded context; the gathered form refuses a wrong shard count
    with pytest.raises(capi.CapError):
        r.post_frame(s, len(cams), cams[-1])
    with pytest.raises(capi.CapError):
        r.post_frame_gathered(s, len(cams), cams[-1], gathered_ptr_of(got), count + 1)
    r.close()


def gathered_ptr_of(_):
    import torch
    return torch.zeros(16, dtype=torch.float32, device="cuda").data_ptr()


def test_gbuffer_feedback_on_shards(native_lib, bluenoise, cornell_path):
    """The reference's default loop (feedback on) over two sharded contexts: per frame both render their tiles with the feedback
    branch, ONE gather brings the chain inputs to the root, the root runs the chain and exports its output + normal/depth, ONE
    broadcast carries them to the other rank (cap_feedback_import).  Bit-identical to the unsharded loop, frame after frame."""
    import torch
    w, h, D, count = 120, 90, 3, 2
    geo = capi.Geometry(cornell_path)
    base = capi.cornell_camera(w, h)
    cams = [base] * 3 + [moved(base, 0.03 * k, 0.0, -0.05 * k) for k in range(1, 3)] + [moved(base, 0.06, 0.0, -0.1)]
    gs = capi.PostSettings()
    flags = capi.RENDER_AOV | capi.RENDER_GBUFFER_FEEDBACK

    def make(idx, n):
        r = capi.Renderer(0)
        r.upload_geometry(geo)
        r.upload_bluenoise(bluenoise)
        r.build_bvh()
        r.set_resolution(w, h)
        r.set_shard(idx, n)
        return r

    ref = make(0, 1)
    want, prev, rays_want = [], cams[0], []
    for f, cam in enumerate(cams):
        ref.set_camera(cam)
        ref.set_prev_camera(prev)
        ref.stats_reset()
        ref.render(f, 1, D, flags)
        ref.post_frame(gs, f, prev)
        want.append(ref.post_readback())
        s = ref.stats()
        rays_want.append((s.rays_primary, s.rays_extension, s.rays_shadow))
        prev = cam
    ref.close()

    ranks = [make(i, count) for i in range(count)]
    prev = cams[0]
    for f, cam in enumerate(cams):
        bufs, rays = [], np.zeros(3, np.int64)
        for r in ranks:
            r.set_camera(cam)
            r.set_prev_camera(prev)
            r.stats_reset()
            r.render(f, 1, D, flags)
            t = torch.empty(r.aov_tile_buffer_floats(), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            r.resolve_aov_tiles(t.data_ptr())
            r.sync()
            s = r.stats()
            rays += (s.rays_primary, s.rays_extension, s.rays_shadow)
            bufs.append(t)
        gathered = torch.cat(bufs)
        torch.cuda.synchronize()
        root = ranks[0]
        root.post_frame_gathered(gs, f, prev, gathered.data_ptr(), count)
        got = root.post_readback()
        assert np.array_equal(bits(got), bits(want[f])), "frame %d: %d pixels differ" % (f, int((bits(got) != bits(want[f])).any(-1).sum()))
        assert tuple(int(x) for x in rays) == rays_want[f]  # the feedback branch ended the same paths on both halves
        fb = torch.empty(root.feedback_buffer_floats(), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        root.feedback_export(fb.data_ptr())
        root.sync()
        for r in ranks[1:]:
            r.feedback_import(fb.data_ptr(), f)  # the broadcast's receiving end
            r.sync()
        prev = cam
    for r in ranks:
        r.close()


def test_sharded_chain_error_paths(native_lib, bluenoise, cornell_path):
    """Call-order and argument errors of the sharded-chain entry points come back as status + message, never as a fault."""
    import torch
    r = capi.Renderer(0)
    r.upload_geometry(capi.Geometry(cornell_path))
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    with pytest.raises(capi.CapError, match="resolution"):
        r.aov_tile_buffer_floats()
    with pytest.raises(capi.CapError, match="resolution"):
        r.feedback_buffer_floats()
    r.set_resolution(64, 48)
    cam = capi.cornell_camera(64, 48)
    r.set_camera(cam)
    r.set_shard(1, 2)
    buf = torch.zeros(max(r.aov_tile_buffer_floats(), r.feedback_buffer_floats()), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with pytest.raises(capi.CapError, match="CAP_RENDER_AOV"):
        r.resolve_aov_tiles(buf.data_ptr())  # nothing rendered yet
    r.render(0, 1, 1)                        # rendered, but without the AOV planes
    with pytest.raises(capi.CapError, match="CAP_RENDER_AOV"):
        r.resolve_aov_tiles(buf.data_ptr())
    with pytest.raises(capi.CapError, match="chain has not run"):
        r.feedback_export(buf.data_ptr())
    with pytest.raises(capi.CapError, match="NULL"):
        r.resolve_aov_tiles(0)
    with pytest.raises(capi.CapError, match="NULL"):
        r.feedback_import(0, 0)
    s = capi.PostSettings(lowres_indirect=1)
    r.set_resolution(63, 48)
    r.set_shard(0, 2)
    with pytest.raises(capi.CapError, match="even width"):
        r.post_frame_gathered(s, 0, cam, buf.data_ptr(), 2)
    r.close()


def _host_threads():
    """The CPU share the test process really has (the GPU box reports 256 hardware threads and grants a cgroup quota of 16)."""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError, AttributeError):
        pass
    return max(1, min(n, 32))


def test_post_chain_parity_1080p(native_lib, bluenoise, cornell_path):
    """The reconstruction chain at the size bench.py times it at (`realtime_frame`, `post_chain`: 1920 x 1080, the reference's window,
    /root/reference/src/viewer/main.cpp:53-54; pipeline raytracing_system.cpp:294-317): depth 1, G-buffer feedback on, default settings,
    two static frames and two moved ones.  At this size post.hip's tile grid is 60 x 135 in eight XCD bands, a stride-7 row-phase tile is a
    fraction of a row, and BlurDisocclusion is sparse -- none of which the 150 x 101 tests reach.  Exact mode: every frame's ray-pass
    planes and chain output bit for bit against the oracle running the same loop (its chain on the host's threads: the passes are
    row-parallel, the images do not depend on the thread count).  `fast_weights`: the same loop on a context of its own inside the
    stated distribution.  Then two frames on three shards through cap_post_frame_gathered, bit-identical to the unsharded chain."""
    import torch
    from oracle import cap_oracle as O
    w, h, D = 1920, 1080, 1
    nt = _host_threads()
    geo = capi.Geometry(cornell_path)

    def make():
        r = capi.Renderer(0)
        r.upload_geometry(geo)
        r.upload_bluenoise(bluenoise)
        r.build_bvh()
        r.set_resolution(w, h)
        return r

    r, rf = make(), make()
    sc = O.Scene(geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes)
    chain = O.PostChain(w, h, threads=nt)
    base = capi.cornell_camera(w, h)
    cams = [base, base, moved(base, 0.02, 0.01, -0.03), moved(base, 0.04, 0.02, -0.06)]
    gs, gf, os_ = capi.PostSettings(), capi.PostSettings(fast_weights=1), O.PostSettings()
    flags = capi.RENDER_AOV | capi.RENDER_GBUFFER_FEEDBACK
    prev, prev_nd, hist = cams[0], np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    for f, cam in enumerate(cams):
        for ctx in (r, rf):
            ctx.set_camera(cam)
            ctx.set_prev_camera(prev)
            ctx.stats_reset()
            ctx.render(f, 1, D, flags)
        ref = sc.render_frame(ocam_of(O, cam), bluenoise, w, h, f, D, threads=nt, feedback=(ocam_of(O, prev), prev_nd, hist))
        for name, kind in (("direct", capi.BUF_DIRECT), ("albedo", capi.BUF_ALBEDO), ("normal_depth", capi.BUF_NORMAL_DEPTH),
                           ("indirect", capi.BUF_INDIRECT)):
            got = r.readback(kind)
            nbad = int((bits(got) != bits(ref[name])).any(-1).sum())
            assert nbad == 0, "frame %d %s: %d pixels differ" % (f, name, nbad)
        s = r.stats()
        assert (s.rays_primary, s.rays_extension, s.rays_shadow) == ref["rays"]
        r.post_frame(gs, f, prev)
        got = r.post_readback()
        want = chain.frame(os_, f, ocam_of(O, cam), ocam_of(O, prev), ref)
        assert np.all(np.isfinite(got))
        nbad = int((bits(got) != bits(want)).any(-1).sum())
        assert nbad == 0, "frame %d chain output: %d pixels differ, max abs %g" % (f, nbad, float(np.abs(got - want).max()))
        # the toleranced mode, carrying its own histories and its own feedback (tolerance: include/capsaicin_hip.h, CapPostSettings)
        rf.post_frame(gf, f, prev)
        fast = rf.post_readback()
        assert np.all(np.isfinite(fast))
        e = np.abs(fast.astype(np.float64) - want)[..., :3] / (np.abs(want[..., :3]) + 1e-3)
        assert np.median(e) <= 2e-5 and np.percentile(e, 99) <= 4e-3 and e.max() <= 3e-2, \
            "fast frame %d: median %.2e p99 %.2e max %.2e" % (f, np.median(e), np.percentile(e, 99), e.max())
        prev, prev_nd, hist = cam, ref["normal_depth"], want
    rf.close()

    # three shards, one gather per frame, the chain on the root: bit-identical to the unsharded chain at this size
    def sequence(count):
        out, prev = [], cams[1]
        r.post_reset()
        for f, cam in enumerate(cams[1:3]):
            r.set_camera(cam)
            if count == 1:
                r.set_shard(0, 1)
                r.render(f, 1, D, capi.RENDER_AOV)
                r.post_frame(gs, f, prev)
            else:
                bufs = []
                for idx in range(count):
                    r.set_shard(idx, count)
                    r.render(f, 1, D, capi.RENDER_AOV)
                    t = torch.empty(r.aov_tile_buffer_floats(), dtype=torch.float32, device="cuda")
                    torch.cuda.synchronize()
                    r.resolve_aov_tiles(t.data_ptr())
                    r.sync()
                    bufs.append(t)
                gathered = torch.cat(bufs)
                torch.cuda.synchronize()
                r.post_frame_gathered(gs, f, prev, gathered.data_ptr(), count)
            out.append(r.post_readback())
            prev = cam
        return out

    one, three = sequence(1), sequence(3)
    for f, (a, b) in enumerate(zip(three, one)):
        assert np.array_equal(bits(a), bits(b)), "sharded frame %d: %d pixels differ" % (f, int((bits(a) != bits(b)).any(-1).sum()))
    r.close()
