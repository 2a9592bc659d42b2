"""The C-ABI library loads on a GPU-less host and exports every function include/*.h declares (no compute calls)."""
import ctypes
import os
import re

from capsaicin_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in ("capsaicin_hip.h", "capsaicin_scene.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(cap_[a-z0-9_]+)\s*\(", text))
    return names


def test_every_declared_symbol_is_exported(native_lib):
    declared = _declared()
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(native_lib, name), "library does not export %s" % name
    # the Python binding covers exactly the declared surface
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)


def test_pod_layouts_match_the_reference():
    # camera_system.h:16-31 (72 B), asset_load_system.h:29-39 (32 B)
    assert ctypes.sizeof(capi.CameraData) == 72
    assert capi.CameraData.focal_length.offset == 12 and capi.CameraData.right.offset == 16
    assert capi.CameraData.forward.offset == 32 and capi.CameraData.up.offset == 48 and capi.CameraData.sensor_size.offset == 64


def test_post_settings_zero_is_the_reference_default(native_lib):
    """a CapPostSettings filled positionally up to lowres_indirect (the round-1 form) leaves the later
    fields zero -- which must be the reference's defaults: USE_VARIANCE on (raytracing_system.h:25), exact weights, kCombined."""
    d = capi.PostSettings()
    assert (d.gather, d.denoise, d.eaw5, d.lowres_indirect) == (1, 1, 1, 0)
    assert (d.eaw_normal_sigma, d.eaw_depth_sigma, d.eaw_luma_sigma) == (128.0, 3.0, 3.0)
    assert (d.gather_normal_sigma, d.gather_depth_sigma, d.gather_luma_sigma) == (64.0, 2.0, 3.0)
    assert abs(d.temporal_upscale_feedback - 0.975) < 1e-7 and abs(d.taa_feedback - 0.9) < 1e-7
    assert (d.disable_variance, d.fast_weights, d.output) == (0, 0, 0) and d.use_variance == 1
    # the fields behind lowres_indirect are the struct's tail: twelve leading fields = 48 bytes
    assert capi.PostSettings.disable_variance.offset == 48 and ctypes.sizeof(capi.PostSettings) == 60


def test_no_gpu_fails_loudly(native_lib):
    if capi.device_count() > 0:
        return
    try:
        capi.Renderer(0)
    except capi.CapError as e:
        assert "cap_ctx_create" in str(e)
    else:
        raise AssertionError("context creation must fail without a HIP device (no CPU fallback)")


def test_product_does_not_touch_the_oracle():
    # the product (package + C ABI sources) never names the oracle directory
    for base, _, files in os.walk(os.path.join(ROOT, "capsaicin_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="ignore").read()
                code = "\n".join(l for l in text.splitlines() if not l.strip().startswith(("//", "#", "*", '"""')))
                assert "cap_oracle" not in code and "oracle/" not in code and "import oracle" not in code, f
