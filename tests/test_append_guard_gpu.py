"""The kernels' queue-append guard (CapStats::guard_append, kernels.hip wave_append2).

A sub-queue's capacity is static because a path keeps the class it got at bounce 0; until round 4 the appends rested on that
argument alone and wrote `slot + class * class_capacity` unchecked.  Here the capacity is made too small on purpose
(cap_debug_set(CAP_DEBUG_QUEUE_CAPACITY_DIV)): the guard must fire, nothing may fault or be written out of bounds (the context
stays usable and the very next normal render is bit-identical to one of a fresh context), and the image must stay finite.
"""
import numpy as np
import pytest

from capsaicin_amd import capi

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def make(cornell_path, bluenoise, w, h, traversal):
    r = capi.Renderer(0)
    r.upload_geometry(capi.Geometry(cornell_path))
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(w, h)
    r.set_camera(capi.cornell_camera(w, h))
    r.set_traversal(traversal)
    return r


@pytest.mark.parametrize("traversal,flags", [(0, 0), (0, capi.RENDER_EXT_MATERIALS), (1, 0), (1, capi.RENDER_EXT_MATERIALS)],
                         ids=["fused", "fused_ext", "tree", "tree_ext"])
def test_guard_fires_and_nothing_is_written_out_of_bounds(native_lib, cornell_path, bluenoise, traversal, flags):
    w, h, spp, depth = 256, 192, 4, 4
    ref_r = make(cornell_path, bluenoise, w, h, traversal)
    r = make(cornell_path, bluenoise, w, h, traversal)
    if flags & capi.RENDER_EXT_MATERIALS:
        import os
        import shutil
        import tempfile
        tmp = tempfile.mkdtemp(prefix="guard_mtl_")
        open(os.path.join(tmp, "c.obj"), "w").write(open(cornell_path).read().replace("mtllib cornellbox.mtl", "mtllib cornell_box.mtl"))
        shutil.copy(os.path.join(os.path.dirname(cornell_path), "cornell_box.mtl"), os.path.join(tmp, "cornell_box.mtl"))
        mats = capi.Geometry(os.path.join(tmp, "c.obj")).materials()
        ref_r.upload_materials(mats)
        r.upload_materials(mats)
    ref_r.render(0, spp, depth, flags)
    ref = ref_r.readback(capi.BUF_ACCUM_SUM)
    ref_stats = ref_r.stats()
    assert ref_stats.guard_append == 0 and ref_stats.guard_shade == 0 and ref_stats.guard_trace_any == 0

    # sub-queues at a quarter of what the classes need: most of bounce 0's appends run past them.  Three quarters of every queue
    # plane then lie BEHIND the last class's sub-queue: a canary word there proves directly that no append left its sub-queue
    # (a first render of the same shape allocates the queues at their final size; the marker goes in after it)
    r.render(0, spp, depth, flags)
    r.accum_reset()
    r.stats_reset()
    r.debug_set(capi.Renderer.DEBUG_QUEUE_CAPACITY_DIV, 4)
    r.debug_set(capi.Renderer.DEBUG_QUEUE_CANARY_FILL, 1)
    r.render(0, spp, depth, flags)
    got = r.readback(capi.BUF_ACCUM_SUM)
    s = r.stats()
    assert s.guard_append > 0, "the append guard did not fire on undersized sub-queues"
    assert r.debug_get(capi.Renderer.DEBUG_QUEUE_CANARY_USED) > 0  # the check sees writes ...
    assert r.debug_get(capi.Renderer.DEBUG_QUEUE_CANARY_BEHIND) == 0, "an append was stored behind the last queue class"
    assert s.guard_shade == 0 and s.guard_trace_any == 0  # no malformed entry was READ either: consumers clamp to the capacity
    assert np.isfinite(got).all() and (got[..., 3] == spp).all()
    # dropped paths lose their indirect light, nothing else: the image is not the reference one ...
    assert (bits(got) != bits(ref)).any()

    # ... and nothing beyond the sub-queues was touched: with the capacity restored the same context renders the reference bits
    r.debug_set(capi.Renderer.DEBUG_QUEUE_CAPACITY_DIV, 1)
    r.accum_reset()
    r.stats_reset()
    r.render(0, spp, depth, flags)
    again = r.readback(capi.BUF_ACCUM_SUM)
    s2 = r.stats()
    assert s2.guard_append == 0
    assert np.array_equal(bits(again), bits(ref))
    assert (s2.rays_primary, s2.rays_extension, s2.rays_shadow) == (ref_stats.rays_primary, ref_stats.rays_extension, ref_stats.rays_shadow)
    r.close()
    ref_r.close()


def test_debug_set_validates(native_lib):
    r = capi.Renderer(0)
    with pytest.raises(capi.CapError):
        r.debug_set(99, 1)
    with pytest.raises(capi.CapError):
        r.debug_set(capi.Renderer.DEBUG_QUEUE_CAPACITY_DIV, 0)
    r.close()
