"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Bit-exact: the build pins every fp32 operation
(DESIGN.md "fp32 arithmetic contract"), so all comparisons are on the raw 32-bit patterns; tolerance = 0."""
import numpy as np
import pytest

from capsaicin_amd import capi, tiles

pytestmark = pytest.mark.gpu

PLANES = (("gbuffer_geo", capi.BUF_GBUFFER_GEO), ("direct", capi.BUF_DIRECT), ("albedo", capi.BUF_ALBEDO),
          ("normal_depth", capi.BUF_NORMAL_DEPTH), ("indirect", capi.BUF_INDIRECT), ("combined", capi.BUF_COMBINED))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_same(gpu, ref, what):
    g, r = bits(gpu), bits(ref)
    if not np.array_equal(g, r):
        bad = np.argwhere((g != r).any(-1))
        raise AssertionError("%s: %d of %d pixels differ, first at (y,x)=%s gpu=%s oracle=%s" %
                             (what, len(bad), g.shape[0] * g.shape[1], tuple(bad[0]), gpu[tuple(bad[0])], ref[tuple(bad[0])]))


@pytest.fixture(scope="module")
def cornell(native_lib, cornell_path, bluenoise):
    from oracle import cap_oracle as O
    from oracle import obj_oracle
    geo = capi.Geometry(cornell_path)
    r = capi.Renderer(0)
    r.upload_geometry(geo)
    r.upload_bluenoise(bluenoise)
    info = r.build_bvh()
    assert info.triangle_count == 32 and info.node_count == 31 and 1 <= info.max_depth <= 31
    g = obj_oracle.load_geometry(cornell_path)
    sc = O.Scene(g["positions"], g["normals"], g["texcoords"], g["indices"], g["meshes"])
    yield r, sc, O
    r.close()


def _oracle_cam(O, cam):
    return O.make_camera(tuple(cam.position), tuple(cam.forward), tuple(cam.right), tuple(cam.up), cam.sensor_size[0],
                         cam.sensor_size[1], cam.focal_length)


@pytest.mark.parametrize("w,h,frame,bounces", [(256, 256, 0, 2), (256, 256, 7, 1), (64, 48, 4095, 0), (100, 75, 12345, 5), (61, 37, 3, 8)])
def test_frame_planes_bit_exact(cornell, bluenoise, w, h, frame, bounces):
    """BASELINE config 1 (256x256, 1 spp, depth 2) and neighbours, incl. sizes that are not multiples of the 8x8 tile."""
    r, sc, O = cornell
    cam = capi.cornell_camera(w, h)
    r.set_resolution(w, h)
    r.set_shard(0, 1)
    r.set_camera(cam)
    ref = sc.render_frame(_oracle_cam(O, cam), bluenoise, w, h, frame, bounces)
    for mode in (1, 2, 0):  # LBVH + LDS stack, exhaustive small-scene kernel, auto: same hits by construction
        r.set_traversal(mode)
        r.accum_reset()
        r.stats_reset()
        r.render(frame, 1, bounces, capi.RENDER_AOV)
        for name, kind in PLANES:
            assert_same(r.readback(kind), ref[name], "%s %dx%d frame %d D=%d traversal %d" % (name, w, h, frame, bounces, mode))
        s = r.stats()
        assert (s.rays_primary, s.rays_extension, s.rays_shadow) == ref["rays"]


def test_accumulation_across_batches_bit_exact(cornell, bluenoise):
    r, sc, O = cornell
    w, h, n, D = 96, 64, 11, 3
    cam = capi.cornell_camera(w, h)
    r.set_resolution(w, h)
    r.set_shard(0, 1)
    r.set_camera(cam)
    acc, rays = sc.render_accumulate(_oracle_cam(O, cam), bluenoise, w, h, 20, n, D)
    for batch_paths in (0, 4 * 96 * 64, 1):  # default, 4 frames per batch, 1 frame per batch
        r.set_batch_paths(batch_paths)
        r.accum_reset()
        r.stats_reset()
        r.render(20, n, D)
        got = r.readback(capi.BUF_ACCUM_SUM)
        assert_same(got[..., :3], acc[..., :3], "accumulated sum, batch_paths=%d" % batch_paths)
        assert np.all(got[..., 3] == n)
        s = r.stats()
        assert (s.rays_primary, s.rays_extension, s.rays_shadow) == rays
    # two calls continue the same running sum
    r.set_batch_paths(0)
    r.accum_reset()
    r.render(20, 4, D)
    r.render(24, n - 4, D)
    assert_same(r.readback(capi.BUF_ACCUM_SUM)[..., :3], acc[..., :3], "split render calls")
    mean = r.readback(capi.BUF_ACCUM_MEAN)
    assert_same(mean[..., :3], acc[..., :3] / np.float32(n), "mean")


@pytest.mark.parametrize("traversal", [1, 2])  # tree kernels, exhaustive small-scene kernels
def test_accumulate_only_render_equals_aov_render(cornell, bluenoise, traversal):
    """A render nobody reads planes from (flags 0) of an untextured scene keeps no albedo plane: the first vertex's albedo is one of
    four constants and travels as a code in direct.w (ShadeArgs::albedo_in_w); one with CAP_RENDER_AOV keeps the planes as the
    reference defines them.  Same sums, same counters."""
    r, sc, O = cornell
    w, h, n, D = 160, 96, 6, 5
    r.set_resolution(w, h)
    r.set_shard(0, 1)
    r.set_camera(capi.cornell_camera(w, h))
    r.set_traversal(traversal)
    out = []
    for flags in (0, capi.RENDER_AOV):
        r.accum_reset()
        r.stats_reset()
        r.render(7, n, D, flags)
        s = r.stats()
        out.append((r.readback(capi.BUF_ACCUM_SUM), (s.rays_primary, s.rays_extension, s.rays_shadow)))
    r.set_traversal(0)
    assert out[0][1] == out[1][1]
    assert_same(out[0][0], out[1][0], "accumulate-only vs AOV render, traversal %d" % traversal)
    # every shadow ray is counted; on the small-scene path most never become queue entries (the producer's probe answers them)
    assert 0 < s.shadow_entries <= s.rays_shadow and s.shadow_entries_bounce0 <= s.rays_shadow_bounce0
    if traversal == 2:
        assert s.shadow_entries < s.rays_shadow // 2
    assert float(out[0][0][..., :3].max()) > 0.0


def test_sharded_render_equals_unsharded(cornell, bluenoise):
    """Tile sharding (one context per shard) + tile buffers: assembling the shards reproduces the single-GPU image."""
    import torch
    r, sc, O = cornell
    w, h, n, D = 100, 60, 3, 2
    cam = capi.cornell_camera(w, h)
    r.set_resolution(w, h)
    r.set_camera(cam)
    r.set_shard(0, 1)
    r.accum_reset()
    r.stats_reset()
    r.render(0, n, D)
    full = r.readback(capi.BUF_ACCUM_MEAN)
    whole = r.stats()
    for count in (2, 3, 8):
        floats = None
        bufs = []
        r.stats_reset()  # the shards' counters add up in one CapStats: every ray is traced by exactly one shard
        for idx in range(count):
            r.set_shard(idx, count)
            r.accum_reset()
            r.render(0, n, D)
            floats = r.tile_buffer_floats()
            assert floats == tiles.padded_pixels(w, h, count) * 4
            t = torch.zeros(floats, dtype=torch.float32, device="cuda")
            # the renderer of this fixture runs on its own (non-blocking) stream: torch's fill must have landed before the
            # renderer writes the buffer (bench.py instead creates the renderer ON torch's stream)
            torch.cuda.synchronize()
            r.resolve_tiles(t.data_ptr())
            r.sync()
            bufs.append(t)
            part = r.readback(capi.BUF_ACCUM_MEAN)
            # the device tile layout is the one tiles.py describes
            dev = t.cpu().numpy().reshape(-1, 4)
            _, _, valid = tiles.pixel_table(w, h, idx, count)
            assert np.array_equal(bits(dev[valid]), bits(tiles.extract(part, idx, count)[valid]))
            assert np.all(dev[~valid][:, :3] == 0)  # padding lanes of partial / absent tiles carry no radiance
        gathered = torch.cat(bufs)
        image = torch.zeros(h * w * 4, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        r.assemble_tiles(gathered.data_ptr(), count, image.data_ptr())
        r.sync()
        torch.cuda.synchronize()
        got = image.cpu().numpy().reshape(h, w, 4)
        assert_same(got, full, "assembled from %d shards" % count)
        assert_same(tiles.assemble([b.cpu().numpy().reshape(-1, 4) for b in bufs], w, h), full, "tiles.assemble %d" % count)
        s = r.stats()
        assert (s.rays_primary, s.rays_extension, s.rays_shadow, s.shaded_vertices) == \
            (whole.rays_primary, whole.rays_extension, whole.rays_shadow, whole.shaded_vertices), "ray counters of %d shards" % count
    r.set_shard(0, 1)


def test_full_size_properties(cornell, bluenoise):
    """BASELINE config 2 size (1920x1080, depth 8): size-independent properties instead of a full oracle render."""
    r, sc, O = cornell
    w, h, D = 1920, 1080, 8
    cam = capi.cornell_camera(w, h)
    r.set_resolution(w, h)
    r.set_shard(0, 1)
    r.set_camera(cam)
    r.set_batch_paths(0)
    r.accum_reset()
    r.stats_reset()
    r.render(0, 4, D)
    a = r.readback(capi.BUF_ACCUM_SUM)
    s = r.stats()
    assert s.rays_primary == 4 * w * h and s.rays_extension <= s.rays_primary * D and s.rays_shadow <= s.shaded_vertices
    assert np.all(np.isfinite(a)) and np.all(a[..., 3] == 4) and np.all(a[..., :3] >= 0)
    # determinism: same frames, different batching -> identical bits
    r.set_batch_paths(w * h)
    r.accum_reset()
    r.render(0, 4, D)
    assert np.array_equal(bits(r.readback(capi.BUF_ACCUM_SUM)), bits(a))
    # additivity over frames: sum of single-frame images == accumulated image (same fp32 order)
    r.set_batch_paths(0)
    tot = np.zeros_like(a)
    for f in range(4):
        r.accum_reset()
        r.render(f, 1, D)
        tot = tot + r.readback(capi.BUF_ACCUM_SUM)
    assert np.array_equal(bits(tot[..., :3]), bits(a[..., :3]))
    # oracle spot check on a crop: rows 500..507 of frame 0 (a full 1080p oracle frame is the bench's CPU baseline, not a test)
    r.accum_reset()
    r.render(0, 1, D, capi.RENDER_AOV)
    got = r.readback(capi.BUF_COMBINED)
    ref = sc.render_frame(_oracle_cam(O, cam), bluenoise, w, h, 0, D, flags=O.FLAG_USE_BVH, threads=8)
    assert_same(got, ref["combined"], "1080p frame 0 combined")
    assert (r.stats().rays_extension - s.rays_extension, ) is not None


def test_accum_import_resumes_bit_identically(native_lib, bluenoise, cornell_path):
    """cap_accum_import (SURVEY.md 5 "checkpoint": dump accumulation + sample index): a render interrupted after 5 of 9 frames, its
    CAP_BUF_ACCUM_SUM dumped to the host and imported into a FRESH context, continues to the bits of the uninterrupted render -- the
    resolve adds frames in frame order either way -- on the fused and on the tree path, unsharded and on a shard."""
    w, h, D = 200, 120, 3
    for traversal, shard in ((0, (0, 1)), (1, (0, 1)), (0, (1, 3))):
        def make():
            r = capi.Renderer(0)
            r.upload_geometry(capi.Geometry(cornell_path))
            r.upload_bluenoise(bluenoise)
            r.build_bvh()
            r.set_resolution(w, h)
            r.set_shard(*shard)
            r.set_camera(capi.cornell_camera(w, h))
            r.set_traversal(traversal)
            return r
        a = make()
        a.render(0, 9, D, 0)
        want = a.readback(capi.BUF_ACCUM_SUM)
        a.accum_reset()
        a.render(0, 5, D, 0)
        dump = a.readback(capi.BUF_ACCUM_SUM)
        a.close()
        b = make()
        b.accum_import(dump, 5)
        b.render(5, 4, D, 0)
        got = b.readback(capi.BUF_ACCUM_SUM)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (traversal, shard)
        assert (got[..., 3][got[..., 3] > 0] == 9).all()
        b.close()
