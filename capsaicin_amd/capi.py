"""ctypes binding of libcapsaicin_hip.so (include/capsaicin_hip.h, include/capsaicin_scene.h).

No rendering happens in Python and nothing here falls back to a CPU path: every call goes through the C ABI and
raises CapError with cap_last_error() when the library reports a failure (e.g. no HIP device).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_PKG, "libcapsaicin_hip.so")
# A/B and diagnostic builds made by tools/build_variant.sh (compile-time switches measured against the product build in one GPU call)
if os.environ.get("CAP_LIB_VARIANT"):
    LIB_PATH = os.path.join(_PKG, "variants", "libcapsaicin_hip_%s.so" % os.environ["CAP_LIB_VARIANT"])

RENDER_AOV = 1
RENDER_EXT_MATERIALS = 2
RENDER_STAGE_TIMERS = 4
RENDER_GBUFFER_FEEDBACK = 8
RENDER_LOWRES_INDIRECT = 16

BUF_GBUFFER_GEO, BUF_DIRECT, BUF_ALBEDO, BUF_NORMAL_DEPTH, BUF_INDIRECT, BUF_COMBINED, BUF_ACCUM_SUM, BUF_ACCUM_MEAN, BUF_INDIRECT_LOWRES = range(9)


class CapError(RuntimeError):
    pass


class CameraData(C.Structure):
    """CameraData, reference src/systems/camera_system.h:16-31 (72 bytes)."""
    _fields_ = [("position", C.c_float * 3), ("focal_length", C.c_float), ("right", C.c_float * 3), ("znear", C.c_float),
                ("forward", C.c_float * 3), ("focus_distance", C.c_float), ("up", C.c_float * 3), ("aperture", C.c_float),
                ("sensor_size", C.c_float * 2)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_extension", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("shaded_vertices", C.c_uint64), ("frames", C.c_uint64), ("ms_total", C.c_double), ("ms_primary", C.c_double),
                ("ms_trace_closest", C.c_double), ("ms_trace_any", C.c_double), ("ms_shade", C.c_double),
                ("ms_resolve", C.c_double), ("launches_trace_closest", C.c_uint64), ("launches_trace_any", C.c_uint64),
                ("launches_shade", C.c_uint64), ("rays_extension_bounce0", C.c_uint64), ("rays_shadow_bounce0", C.c_uint64),
                ("guard_shade", C.c_uint64), ("guard_trace_any", C.c_uint64), ("guard_last", C.c_uint64), ("ms_post", C.c_double),
                ("post_frames", C.c_uint64), ("ms_direct", C.c_double), ("ms_post_pass", C.c_double * 5), ("shadow_entries", C.c_uint64),
                ("shadow_entries_bounce0", C.c_uint64), ("guard_append", C.c_uint64), ("lane1_dropped", C.c_uint64)]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n == "ms_post_pass" else getattr(self, n)) for n, _ in self._fields_}


class BvhInfo(C.Structure):
    _fields_ = [("triangle_count", C.c_uint32), ("node_count", C.c_uint32), ("max_depth", C.c_uint32),
                ("stack_entries", C.c_uint32), ("bounds_lo", C.c_float * 3), ("bounds_hi", C.c_float * 3),
                ("build_ms", C.c_double)]


OUTPUT_COMBINED, OUTPUT_DIRECT, OUTPUT_INDIRECT, OUTPUT_VARIANCE = range(4)  # SettingsComponent::output, gui_system.h:11-17


class PostSettings(C.Structure):
    """SettingsComponent fields read by the reconstruction chain, reference src/systems/gui_system.h:20-37.  Every field behind
    lowres_indirect reads 0 as the reference default (so `use_variance`, default true, travels as disable_variance)."""
    _fields_ = [("gather", C.c_int32), ("denoise", C.c_int32), ("eaw5", C.c_int32), ("eaw_normal_sigma", C.c_float),
                ("eaw_depth_sigma", C.c_float), ("eaw_luma_sigma", C.c_float), ("gather_normal_sigma", C.c_float),
                ("gather_depth_sigma", C.c_float), ("gather_luma_sigma", C.c_float), ("temporal_upscale_feedback", C.c_float),
                ("taa_feedback", C.c_float), ("lowres_indirect", C.c_int32), ("disable_variance", C.c_int32), ("fast_weights", C.c_int32),
                ("output", C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        lib().cap_post_settings_default(C.byref(self))  # the reference's defaults, from the library itself
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def use_variance(self):  # RaytracingOptions::use_variance, raytracing_system.h:25
        return 0 if self.disable_variance else 1

    @use_variance.setter
    def use_variance(self, v):
        self.disable_variance = 0 if v else 1


class GeometryView(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("texcoords", C.POINTER(C.c_float)),
                ("indices", C.POINTER(C.c_uint32)), ("meshes", C.POINTER(C.c_uint32)), ("vertex_count", C.c_uint32),
                ("index_count", C.c_uint32), ("mesh_count", C.c_uint32), ("texture_count", C.c_uint32),
                ("material_count", C.c_uint32)]


# every symbol include/capsaicin_hip.h and include/capsaicin_scene.h declare: (restype, argtypes)
_vp, _u32, _u64, _i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
SYMBOLS = {
    "cap_last_error": (C.c_char_p, []),
    "cap_device_count": (_i, []),
    "cap_ctx_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "cap_ctx_destroy": (None, [_vp]),
    "cap_scene_upload": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32]),
    "cap_texture_upload": (_i, [_vp, _u32, _vp, _u32, _u32]),
    "cap_bluenoise_upload": (_i, [_vp, _vp]),
    "cap_materials_upload": (_i, [_vp, _vp, _u32]),
    "cap_bvh_build": (_i, [_vp]),
    "cap_set_bvh_build": (_i, [_vp, _u32]),
    "cap_bvh_info": (_i, [_vp, C.POINTER(BvhInfo)]),
    "cap_bvh_readback": (_i, [_vp, _vp, _vp]),
    "cap_bvh_wide_readback": (_i, [_vp, _vp, _vp, _vp]),
    "cap_camera_set": (_i, [_vp, C.POINTER(CameraData)]),
    "cap_prev_camera_set": (_i, [_vp, C.POINTER(CameraData)]),
    "cap_set_resolution": (_i, [_vp, _u32, _u32]),
    "cap_set_shard": (_i, [_vp, _u32, _u32]),
    "cap_set_batch_paths": (_i, [_vp, _u64]),
    "cap_set_traversal": (_i, [_vp, _u32]),
    "cap_debug_set": (_i, [_vp, _u32, _u64]),
    "cap_debug_get": (_i, [_vp, _u32, C.POINTER(_u64)]),
    "cap_debug_switch_index": (_i, [C.c_char_p]),
    "cap_render": (_i, [_vp, _u32, _u32, _u32, _u32]),
    "cap_accum_reset": (_i, [_vp]),
    "cap_accum_import": (_i, [_vp, _vp, _u64]),
    "cap_sync": (_i, [_vp]),
    "cap_readback": (_i, [_vp, _i, _vp]),
    "cap_stats_get": (_i, [_vp, C.POINTER(Stats)]),
    "cap_stats_reset": (_i, [_vp]),
    "cap_tile_buffer_floats": (_i, [_vp, C.POINTER(C.c_size_t)]),
    "cap_resolve_tiles": (_i, [_vp, _vp]),
    "cap_assemble_tiles": (_i, [_vp, _vp, _u32, _vp]),
    "cap_post_settings_default": (None, [C.POINTER(PostSettings)]),
    "cap_post_frame": (_i, [_vp, C.POINTER(PostSettings), _u32, C.POINTER(CameraData)]),
    "cap_aov_tile_buffer_floats": (_i, [_vp, C.POINTER(C.c_size_t)]),
    "cap_resolve_aov_tiles": (_i, [_vp, _vp]),
    "cap_post_frame_gathered": (_i, [_vp, C.POINTER(PostSettings), _u32, C.POINTER(CameraData), _vp, _u32]),
    "cap_feedback_buffer_floats": (_i, [_vp, C.POINTER(C.c_size_t)]),
    "cap_feedback_export": (_i, [_vp, _vp]),
    "cap_feedback_import": (_i, [_vp, _vp, _u32]),
    "cap_post_reset": (_i, [_vp]),
    "cap_post_readback": (_i, [_vp, _vp]),
    "cap_obj_load": (_i, [C.c_char_p, C.c_char_p, C.POINTER(_vp)]),
    "cap_geometry_free": (None, [_vp]),
    "cap_obj_set_threads": (None, [C.c_int]),
    "cap_geometry_view": (_i, [_vp, C.POINTER(GeometryView)]),
    "cap_geometry_texture_name": (C.c_char_p, [_vp, _u32]),
    "cap_geometry_warning": (C.c_char_p, [_vp]),
    "cap_geometry_materials": (_i, [_vp, _vp]),
    "cap_scene_upload_geometry": (_i, [_vp, _vp]),
    "cap_comm_unique_id": (_i, [_vp]),
    "cap_comm_init_rank": (_i, [_vp, _vp, _u32, _u32]),
    "cap_comm_gather_frame": (_i, [_vp]),
    "cap_comm_init_all": (_i, [_vp, _u32]),
    "cap_comm_gather_frame_all": (_i, [_vp, _u32]),
    "cap_comm_image": (_i, [_vp, C.POINTER(C.c_void_p)]),
    "cap_comm_readback": (_i, [_vp, _vp]),
    "cap_comm_info": (_i, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "cap_comm_destroy": (_i, [_vp]),
    "cap_comm_abort": (_i, [_vp]),
    "cap_image_decode": (_i, [_vp, C.c_size_t, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(_u32), C.POINTER(_u32)]),
    "cap_image_free": (None, [_vp]),
    "cap_host_sah_build": (_i, [_vp, _u32, _vp, _vp, C.POINTER(_u32)]),
    "cap_host_wide_build": (_i, [_vp, _u32, _vp, _vp, _vp, _u32, _vp, _vp]),
}

_LIB = None


def build_native(force=False):
    """Compile the HIP library in-tree with hipcc for gfx950 (capsaicin_amd/csrc/Makefile)."""
    args = ["make", "-C", os.path.join(_PKG, "csrc"), "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise CapError("native library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % LIB_PATH)
        # PyTorch-ROCm bundles its own copy of the HIP runtime (same SONAME).  A process must hold exactly one runtime: if torch
        # is going to be used next to this library (device buffers for the tile gather, torch.distributed), its copy has to be
        # the one already loaded when libcapsaicin_hip.so resolves libamdhip64.so.7; the other order leaves torch with
        # "No HIP GPUs are available".  torch is plumbing only; nothing below needs it.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch-less host: the system runtime is used
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _LIB = L
    return _LIB


def _check(rc, what):
    if rc != 0:
        raise CapError("%s failed (status %d): %s" % (what, rc, lib().cap_last_error().decode()))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    return int(lib().cap_device_count())


def load_bluenoise(path=None):
    """assets/bluenoise256.rgba: raw RGBA8 texels of the reference's blue-noise texture (the sampler's RNG)."""
    path = path or os.path.join(_ROOT, "assets", "bluenoise256.rgba")
    return np.fromfile(path, np.uint8).reshape(256, 256, 4)


_SCENE_CONFIG = None


def scene_config():
    """assets/scene_config.json: the scene constants of the benchmark workloads that the reference does not fix."""
    global _SCENE_CONFIG
    if _SCENE_CONFIG is None:
        import json
        cfg = json.load(open(os.path.join(os.path.dirname(_PKG), "assets", "scene_config.json")))
        cfg["cornell_ggx"] = {k: v for k, v in cfg["cornell_ggx"].items() if not k.startswith("_")}
        _SCENE_CONFIG = cfg
    return _SCENE_CONFIG


def camera_from_config(c, width, height):
    """CameraData from a config entry; right / up by the reference's own formulas when absent (input_system.cpp:134-141)."""
    cam = CameraData()
    f = np.float64(c["forward"])
    f /= np.linalg.norm(f)
    right = np.float64(c["right"]) if "right" in c else -np.cross(f, (0.0, 1.0, 0.0))
    right /= np.linalg.norm(right)
    up = np.float64(c["up"]) if "up" in c else np.cross(f, right)
    cam.position[:] = c["position"]
    cam.forward[:] = f
    cam.right[:] = right
    cam.up[:] = up
    cam.focal_length = c["focal_length"]
    cam.sensor_size[0] = c["sensor_x"]
    cam.sensor_size[1] = np.float32(c["sensor_x"]) * (np.float32(height) / np.float32(width))  # camera_system.cpp:10-17
    return cam


def cornell_camera(width, height):
    """The Cornell-box view fixed by SURVEY.md 8d (assets/scene_config.json)."""
    return camera_from_config(scene_config()["cornell_camera"], width, height)


class Geometry:
    """Host-side GeometryStorage produced by the native OBJ loader (cap_obj_load)."""

    def __init__(self, obj_path, mtl_dir=None):
        self.h = C.c_void_p()
        _check(lib().cap_obj_load(obj_path.encode(), (mtl_dir or "").encode(), C.byref(self.h)), "cap_obj_load")
        v = GeometryView()
        _check(lib().cap_geometry_view(self.h, C.byref(v)), "cap_geometry_view")
        self.view = v
        nv, ni, nm = v.vertex_count, v.index_count, v.mesh_count
        self.positions = np.ctypeslib.as_array(v.positions, (3 * nv,)).copy() if nv else np.zeros(0, np.float32)
        self.normals = np.ctypeslib.as_array(v.normals, (3 * nv,)).copy() if nv else np.zeros(0, np.float32)
        self.texcoords = np.ctypeslib.as_array(v.texcoords, (2 * nv,)).copy() if nv else np.zeros(0, np.float32)
        self.indices = np.ctypeslib.as_array(v.indices, (ni,)).copy() if ni else np.zeros(0, np.uint32)
        self.meshes = np.ctypeslib.as_array(v.meshes, (nm * 8,)).copy().reshape(-1, 8) if nm else np.zeros((0, 8), np.uint32)
        self.texture_names = [lib().cap_geometry_texture_name(self.h, i).decode() for i in range(v.texture_count)]
        self.material_count = v.material_count
        self.warning = lib().cap_geometry_warning(self.h).decode()

    def materials(self):
        m = np.zeros((self.meshes.shape[0], 12), np.float32)
        _check(lib().cap_geometry_materials(self.h, _p(m)), "cap_geometry_materials")
        return m

    def __del__(self):
        if getattr(self, "h", None):
            lib().cap_geometry_free(self.h)
            self.h = None


def host_sah_build(tri_lo, tri_hi):
    """The host-side SAH tree over triangle boxes (no GPU): (nodes [n-1, 16] float32, order [n] uint32, depth)."""
    lo, hi = np.asarray(tri_lo, np.float32), np.asarray(tri_hi, np.float32)
    n = len(lo)
    boxes = np.zeros((n, 8), np.float32)
    boxes[:, 0:3], boxes[:, 4:7] = lo, hi
    nodes = np.zeros((max(n - 1, 1), 16), np.float32)
    order = np.zeros(n, np.uint32)
    depth = C.c_uint32()
    _check(lib().cap_host_sah_build(_p(boxes), n, _p(nodes), _p(order), C.byref(depth)), "cap_host_sah_build")
    return nodes[:max(n - 1, 0)], order, int(depth.value)


def image_decode(data, name=None):
    """Texture file bytes -> [h, w, 4] uint8 (PNG / TGA / binary PPM), as TextureSystem's stbi_load(..., 4) would hand over."""
    buf = np.frombuffer(bytes(data), np.uint8)
    px, w, h = C.c_void_p(), _u32(), _u32()
    _check(lib().cap_image_decode(_p(buf), buf.size, name.encode() if name else None, C.byref(px), C.byref(w), C.byref(h)), "cap_image_decode")
    out = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint8)), (h.value, w.value, 4)).copy()
    lib().cap_image_free(px)
    return out


def host_wide_build(nodes, n, scene_lo, scene_hi):
    """Binary tree (host_sah_build's nodes) -> compressed 8-wide view: (wide nodes [count, 20] uint32, tri_src, depth, top)."""
    nodes = np.ascontiguousarray(nodes, np.float32)
    lo, hi = np.ascontiguousarray(scene_lo, np.float32), np.ascontiguousarray(scene_hi, np.float32)
    wide = np.zeros((max(n, 1), 20), np.uint32)
    src = np.zeros(max(n, 1), np.uint32)
    info = np.zeros(3, np.uint32)
    _check(lib().cap_host_wide_build(_p(nodes) if n > 1 else None, n, _p(lo), _p(hi), _p(wide), wide.shape[0], _p(src), _p(info)),
           "cap_host_wide_build")
    return wide[:int(info[0])], src[:n], int(info[1]), int(info[2])


def comm_unique_id():
    """128-byte RCCL id made by rank 0; the caller carries it to the other ranks."""
    buf = (C.c_uint8 * 128)()
    _check(lib().cap_comm_unique_id(C.cast(buf, C.c_void_p)), "cap_comm_unique_id")
    return bytes(buf)


def comm_init_all(renderers):
    arr = (C.c_void_p * len(renderers))(*[r.ctx for r in renderers])
    _check(lib().cap_comm_init_all(C.cast(arr, C.c_void_p), len(renderers)), "cap_comm_init_all")


def comm_gather_frame_all(renderers):
    arr = (C.c_void_p * len(renderers))(*[r.ctx for r in renderers])
    _check(lib().cap_comm_gather_frame_all(C.cast(arr, C.c_void_p), len(renderers)), "cap_comm_gather_frame_all")


class Renderer:
    """One CapContext (= one GPU).  Mirrors what RaytracingSystem::Run consumes and produces."""

    def __init__(self, device=0, stream=None):
        self.ctx = C.c_void_p()
        _check(lib().cap_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self.ctx)), "cap_ctx_create")
        self.width = self.height = 0

    def close(self):
        if getattr(self, "ctx", None):
            lib().cap_ctx_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    # ---- scene ----
    def upload_scene(self, positions, normals, texcoords, indices, meshes):
        a = [np.ascontiguousarray(positions, np.float32).ravel(), np.ascontiguousarray(normals, np.float32).ravel(),
             np.ascontiguousarray(texcoords, np.float32).ravel(), np.ascontiguousarray(indices, np.uint32).ravel(),
             np.ascontiguousarray(meshes, np.uint32).reshape(-1, 8)]
        _check(lib().cap_scene_upload(self.ctx, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), a[0].size // 3, a[3].size,
                                      a[4].shape[0]), "cap_scene_upload")

    def upload_geometry(self, geo):
        _check(lib().cap_scene_upload_geometry(self.ctx, geo.h), "cap_scene_upload_geometry")

    def upload_texture(self, index, rgba8):
        if rgba8 is None:
            _check(lib().cap_texture_upload(self.ctx, index, None, 0, 0), "cap_texture_upload")
            return
        t = np.ascontiguousarray(rgba8, np.uint8)
        _check(lib().cap_texture_upload(self.ctx, index, _p(t), t.shape[1], t.shape[0]), "cap_texture_upload")

    def upload_bluenoise(self, rgba8):
        t = np.ascontiguousarray(rgba8, np.uint8)
        assert t.size == 256 * 256 * 4
        _check(lib().cap_bluenoise_upload(self.ctx, _p(t)), "cap_bluenoise_upload")

    def upload_materials(self, materials):
        m = np.ascontiguousarray(materials, np.float32).reshape(-1, 12)
        _check(lib().cap_materials_upload(self.ctx, _p(m), m.shape[0]), "cap_materials_upload")

    BVH_BUILD_AUTO, BVH_BUILD_LBVH, BVH_BUILD_SAH, BVH_BUILD_PLOC, BVH_BUILD_SAH_DEVICE = 0, 1, 2, 3, 4  # CapBvhBuild

    def set_bvh_build(self, mode):
        """0 auto (clustering build on the device above 64 triangles), 1 Morton hierarchy on the device, 2 SAH on the host, 3 clustering,
        4 SAH on the device (surface-area splits down to small segments, clustering inside)."""
        _check(lib().cap_set_bvh_build(self.ctx, mode), "cap_set_bvh_build")

    def build_bvh(self):
        _check(lib().cap_bvh_build(self.ctx), "cap_bvh_build")
        return self.bvh_info()

    def bvh_info(self):
        bi = BvhInfo()
        _check(lib().cap_bvh_info(self.ctx, C.byref(bi)), "cap_bvh_info")
        return bi

    def bvh_readback(self):
        bi = self.bvh_info()
        nodes = np.zeros((bi.node_count, 16), np.float32)
        leaves = np.zeros(bi.triangle_count, np.uint32)
        _check(lib().cap_bvh_readback(self.ctx, _p(nodes), _p(leaves)), "cap_bvh_readback")
        return nodes, leaves

    # ---- view ----
    def bvh_wide_readback(self):
        """(wide nodes [count, 20] uint32, tri_src [triangles] uint32, depth, top) of the compressed 8-wide view."""
        info = np.zeros(3, np.uint32)
        _check(lib().cap_bvh_wide_readback(self.ctx, None, None, _p(info)), "cap_bvh_wide_readback")
        nodes = np.zeros((int(info[0]), 20), np.uint32)
        src = np.zeros(max(1, self.bvh_info().triangle_count), np.uint32)
        _check(lib().cap_bvh_wide_readback(self.ctx, _p(nodes), _p(src), _p(info)), "cap_bvh_wide_readback")
        return nodes, src[:self.bvh_info().triangle_count], int(info[1]), int(info[2])

    def bvh_wide_info(self):
        """(wide nodes, depth, leading top-level nodes) of the compressed 8-wide view, without reading it back."""
        info = np.zeros(3, np.uint32)
        _check(lib().cap_bvh_wide_readback(self.ctx, None, None, _p(info)), "cap_bvh_wide_readback")
        return int(info[0]), int(info[1]), int(info[2])

    def set_camera(self, cam):
        _check(lib().cap_camera_set(self.ctx, C.byref(cam)), "cap_camera_set")

    def set_prev_camera(self, cam):
        _check(lib().cap_prev_camera_set(self.ctx, C.byref(cam)), "cap_prev_camera_set")

    def set_resolution(self, width, height):
        _check(lib().cap_set_resolution(self.ctx, width, height), "cap_set_resolution")
        self.width, self.height = width, height

    def set_shard(self, index, count):
        _check(lib().cap_set_shard(self.ctx, index, count), "cap_set_shard")

    def set_traversal(self, mode):
        """0 auto, 1 LBVH + LDS stack, 2 exhaustive (small scenes)."""
        _check(lib().cap_set_traversal(self.ctx, mode), "cap_set_traversal")

    DEBUG_QUEUE_CAPACITY_DIV, DEBUG_WIDE_DEPTH_LIMIT, DEBUG_WIDE_IN_USE, DEBUG_FAIL_LANE1, DEBUG_LANES_USED = 1, 2, 3, 4, 5
    DEBUG_QUEUE_CANARY_FILL, DEBUG_QUEUE_CANARY_BEHIND, DEBUG_QUEUE_CANARY_USED, DEBUG_SELFTEST_DIV, DEBUG_NEE_PAIRS = 6, 7, 8, 9, 10

    def debug_set(self, key, value):
        _check(lib().cap_debug_set(self.ctx, key, value), "cap_debug_set")

    DEBUG_SWITCH_BASE = 64

    def debug_switch(self, name, value):
        """Sets one of the A/B switches of the context's table by its (environment-variable) name; value None = the product's choice."""
        i = lib().cap_debug_switch_index(name.encode())
        if i < 0:
            raise CapError("unknown switch %s" % name)
        self.debug_set(self.DEBUG_SWITCH_BASE + i, 0xFFFFFFFFFFFFFFFF if value is None else int(value))

    def debug_get(self, key):
        v = _u64()
        _check(lib().cap_debug_get(self.ctx, key, C.byref(v)), "cap_debug_get")
        return int(v.value)

    def set_batch_paths(self, n):
        _check(lib().cap_set_batch_paths(self.ctx, n), "cap_set_batch_paths")

    # ---- render ----
    def render(self, frame_begin, n_frames, num_bounces, flags=0):
        _check(lib().cap_render(self.ctx, frame_begin, n_frames, num_bounces, flags), "cap_render")

    def accum_reset(self):
        _check(lib().cap_accum_reset(self.ctx), "cap_accum_reset")

    def accum_import(self, sum_rgba, frames):
        """Continues a dumped accumulation: sum_rgba = readback(BUF_ACCUM_SUM) of the interrupted render, frames = its frame count."""
        a = np.ascontiguousarray(sum_rgba, np.float32)
        assert a.shape == (self.height, self.width, 4)
        _check(lib().cap_accum_import(self.ctx, _p(a), int(frames)), "cap_accum_import")

    def sync(self):
        _check(lib().cap_sync(self.ctx), "cap_sync")

    def readback(self, kind):
        if kind == BUF_INDIRECT_LOWRES:
            out = np.zeros((self.height // 2, self.width // 2, 4), np.float32)
        else:
            out = np.zeros((self.height, self.width, 4), np.float32)
        _check(lib().cap_readback(self.ctx, kind, _p(out)), "cap_readback")
        return out

    def stats(self):
        s = Stats()
        _check(lib().cap_stats_get(self.ctx, C.byref(s)), "cap_stats_get")
        return s

    def stats_reset(self):
        _check(lib().cap_stats_reset(self.ctx), "cap_stats_reset")

    # ---- reconstruction chain ----
    def post_frame(self, settings, frame_count, prev_camera):
        """Gather -> Accumulate -> Denoise -> Combine -> TAA on the last CAP_RENDER_AOV frame (raytracing_system.cpp:294-317)."""
        _check(lib().cap_post_frame(self.ctx, C.byref(settings), frame_count, C.byref(prev_camera)), "cap_post_frame")

    def post_reset(self):
        _check(lib().cap_post_reset(self.ctx), "cap_post_reset")

    def post_readback(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        _check(lib().cap_post_readback(self.ctx, _p(out)), "cap_post_readback")
        return out

    # ---- multi-GPU tile exchange ----
    def tile_buffer_floats(self):
        n = C.c_size_t()
        _check(lib().cap_tile_buffer_floats(self.ctx, C.byref(n)), "cap_tile_buffer_floats")
        return int(n.value)

    def aov_tile_buffer_floats(self):
        n = C.c_size_t()
        _check(lib().cap_aov_tile_buffer_floats(self.ctx, C.byref(n)), "cap_aov_tile_buffer_floats")
        return n.value

    def resolve_aov_tiles(self, device_ptr):
        _check(lib().cap_resolve_aov_tiles(self.ctx, C.c_void_p(device_ptr)), "cap_resolve_aov_tiles")

    def post_frame_gathered(self, settings, frame_count, prev_camera, device_gathered, shard_count):
        _check(lib().cap_post_frame_gathered(self.ctx, C.byref(settings), frame_count, C.byref(prev_camera), C.c_void_p(device_gathered),
                                             shard_count), "cap_post_frame_gathered")

    def feedback_buffer_floats(self):
        n = C.c_size_t()
        _check(lib().cap_feedback_buffer_floats(self.ctx, C.byref(n)), "cap_feedback_buffer_floats")
        return n.value

    def feedback_export(self, device_ptr):
        _check(lib().cap_feedback_export(self.ctx, C.c_void_p(device_ptr)), "cap_feedback_export")

    def feedback_import(self, device_ptr, frame_count):
        _check(lib().cap_feedback_import(self.ctx, C.c_void_p(device_ptr), frame_count), "cap_feedback_import")

    def resolve_tiles(self, device_ptr):
        _check(lib().cap_resolve_tiles(self.ctx, C.c_void_p(device_ptr)), "cap_resolve_tiles")

    def assemble_tiles(self, device_src, shard_count, device_image):
        _check(lib().cap_assemble_tiles(self.ctx, C.c_void_p(device_src), shard_count, C.c_void_p(device_image)), "cap_assemble_tiles")

    # ---- the exchange below Python: RCCL gather of tile radiance + assembly on rank 0 (cap_comm_*) ----
    def comm_init_rank(self, unique_id, rank, nranks):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _check(lib().cap_comm_init_rank(self.ctx, C.cast(buf, C.c_void_p), rank, nranks), "cap_comm_init_rank")

    def comm_gather_frame(self):
        _check(lib().cap_comm_gather_frame(self.ctx), "cap_comm_gather_frame")

    def comm_image_ptr(self):
        p = C.c_void_p()
        _check(lib().cap_comm_image(self.ctx, C.byref(p)), "cap_comm_image")
        return p.value

    def comm_readback(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        _check(lib().cap_comm_readback(self.ctx, _p(out)), "cap_comm_readback")
        return out

    def comm_info(self):
        r, n, u = _u32(), _u32(), _u32()
        _check(lib().cap_comm_info(self.ctx, C.byref(r), C.byref(n), C.byref(u)), "cap_comm_info")
        return int(r.value), int(n.value), bool(u.value)

    def comm_destroy(self):
        _check(lib().cap_comm_destroy(self.ctx), "cap_comm_destroy")

    def comm_abort(self):
        _check(lib().cap_comm_abort(self.ctx), "cap_comm_abort")
