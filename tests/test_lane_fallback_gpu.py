"""cap_render's second batch lane (tree path: consecutive batches alternate between two working sets on two streams) and its
fallback: when the second working set cannot be allocated the call runs on one lane -- same bits --, says so in
CapStats::lane1_dropped, and does not try the failed size again until the needed size changes.  The failure is
provoked with cap_debug_set(CAP_DEBUG_FAIL_LANE1)."""
import os
import sys

import numpy as np
import pytest

from capsaicin_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_one_lane_fallback_is_bit_identical_and_counted(native_lib, bluenoise):
    import make_sponza_class as gen
    pos, nrm, uv, idx, meshes, texs = gen.arrays(0.1, 64)
    w, h, spp, D = 96, 64, 4, 3
    r = capi.Renderer(0)
    r.upload_scene(pos, nrm, uv, idx, meshes)
    for i, t in enumerate(texs):
        r.upload_texture(i, t)
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(w, h)
    r.set_camera(capi.camera_from_config(dict(gen.camera(), sensor_x=0.036), w, h))
    r.set_batch_paths(w * h)  # one frame per batch: four batches, alternating lanes

    def render():
        r.accum_reset()
        r.stats_reset()
        r.render(0, spp, D, 0)
        return r.readback(capi.BUF_ACCUM_SUM), r.stats(), r.debug_get(capi.Renderer.DEBUG_LANES_USED)

    ref, s, lanes = render()
    if os.environ.get("CAP_NO_TWO_LANES"):
        pytest.skip("CAP_NO_TWO_LANES is set: nothing to fall back from")
    assert lanes == 2 and s.lane1_dropped == 0
    r.debug_set(capi.Renderer.DEBUG_FAIL_LANE1, 1)
    a, s1, lanes = render()
    assert lanes == 1 and s1.lane1_dropped == 1
    assert np.array_equal(bits(a), bits(ref))
    assert (s1.rays_primary, s1.rays_extension, s1.rays_shadow) == (s.rays_primary, s.rays_extension, s.rays_shadow)
    # the failed size is remembered: the next call goes straight to one lane (and counts itself)
    a, s2, lanes = render()
    assert lanes == 1 and s2.lane1_dropped == 1 and np.array_equal(bits(a), bits(ref))
    # "memory has been released": two lanes again, same bits
    r.debug_set(capi.Renderer.DEBUG_FAIL_LANE1, 0)
    a, s3, lanes = render()
    assert lanes == 2 and s3.lane1_dropped == 0 and np.array_equal(bits(a), bits(ref))
    r.close()
