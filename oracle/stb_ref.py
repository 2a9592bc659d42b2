"""Test infrastructure only: ctypes access to oracle/_ref/libstb_ref.so, the reference's own texture decoder.

The reference loads every texture with stbi_load(file, &w, &h, &n, 4) (src/core/src/systems/texture_system.cpp:41-45) from the
header it vendors, src/core/src/utils/stb_image.h (v2.25).  That header is plain C with no dependencies, so `make -C oracle ref`
compiles it where it lies under /root/reference into oracle/_ref/ (git-ignored; it travels to the GPU box as a built file).
Nothing under capsaicin_amd/ loads it: only tests/ and tools/make_image_fixtures.py do, to pin capsaicin_amd/csrc/image_decode.cpp
and jpeg_decode.cpp against the decoder the reference actually runs.
"""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libstb_ref.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_PATH)
        _lib.stbi_load_from_memory.restype = C.c_void_p
        _lib.stbi_load_from_memory.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        _lib.stbi_image_free.argtypes = [C.c_void_p]
        _lib.stbi_failure_reason.restype = C.c_char_p
    return _lib


def decode(data):
    """stbi_load_from_memory(data, 4 channels) -> (h, w, 4) uint8, or None when stb refuses the file."""
    lib = _load()
    data = bytes(data)
    w, h, n = C.c_int(), C.c_int(), C.c_int()
    p = lib.stbi_load_from_memory(data, len(data), C.byref(w), C.byref(h), C.byref(n), 4)
    if not p:
        return None
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value, 4)).copy()
    lib.stbi_image_free(p)
    return out


def failure_reason():
    r = _load().stbi_failure_reason()
    return r.decode() if r else ""
