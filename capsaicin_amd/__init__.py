"""capsaicin_amd — MI355X-native wavefront path tracer behind capsaicin's scene / camera / ray-pass surface.

The product is the HIP library capsaicin_amd/libcapsaicin_hip.so (C ABI in include/capsaicin_hip.h and
include/capsaicin_scene.h, C++ host API in capsaicin_amd/csrc/capsaicin.h).  This Python package is only the
thin ctypes plumbing tests and bench.py use to reach that C ABI; it contains no rendering code and no CPU
fallback: if the library is missing, importing `capi` raises.
"""
from . import capi  # noqa: F401
from .capi import Renderer, Geometry, CameraData, cornell_camera, load_bluenoise, build_native  # noqa: F401

__all__ = ["capi", "Renderer", "Geometry", "CameraData", "cornell_camera", "load_bluenoise", "build_native"]
