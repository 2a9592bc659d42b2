"""ctypes binding of oracle/libcap_oracle.so (ORACLE — test infrastructure only, see cap_oracle.h)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("focal_length", C.c_float), ("right", C.c_float * 3), ("znear", C.c_float),
                ("forward", C.c_float * 3), ("focus_distance", C.c_float), ("up", C.c_float * 3), ("aperture", C.c_float),
                ("sensor_size", C.c_float * 2)]


class Texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("normals", C.c_void_p), ("texcoords", C.c_void_p), ("indices", C.c_void_p),
                ("meshes", C.c_void_p), ("mesh_count", C.c_uint32), ("vertex_count", C.c_uint32), ("index_count", C.c_uint32),
                ("textures", C.c_void_p), ("texture_count", C.c_uint32), ("materials", C.c_void_p)]


class FrameOutputs(C.Structure):
    _fields_ = [("gbuffer_geo", C.c_void_p), ("direct", C.c_void_p), ("albedo", C.c_void_p), ("normal_depth", C.c_void_p),
                ("indirect", C.c_void_p), ("combined", C.c_void_p), ("rays", C.c_uint64 * 3), ("indirect_lowres", C.c_void_p)]


class PostSettings(C.Structure):
    """SettingsComponent subset with the reference defaults (gui_system.h:20-37)."""
    _fields_ = [("gather", C.c_int), ("denoise", C.c_int), ("eaw5", C.c_int), ("eaw_normal_sigma", C.c_float),
                ("eaw_depth_sigma", C.c_float), ("eaw_luma_sigma", C.c_float), ("gather_normal_sigma", C.c_float),
                ("gather_depth_sigma", C.c_float), ("gather_luma_sigma", C.c_float), ("temporal_upscale_feedback", C.c_float),
                ("taa_feedback", C.c_float), ("lowres_indirect", C.c_int), ("use_variance", C.c_int), ("output", C.c_int)]

    def __init__(self, **kw):
        super().__init__(1, 1, 1, 128.0, 3.0, 3.0, 64.0, 2.0, 3.0, 0.975, 0.9, 0, 1, 0)
        for k, v in kw.items():
            setattr(self, k, v)


FLAG_USE_BVH = 1
FLAG_EXT_MATERIALS = 2
FLAG_LOWRES_INDIRECT = 4


def build(force=False):
    so = os.path.join(_HERE, "libcap_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("cap_oracle.cpp", "cap_oracle_post.cpp", "cap_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcap_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_scene_create.restype = C.c_void_p
        L.oracle_scene_create.argtypes = [C.POINTER(SceneDesc)]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_render_frame.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_void_p] + [C.c_uint32] * 6 + [C.POINTER(FrameOutputs)]
        L.oracle_render_frame_rows.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_void_p] + [C.c_uint32] * 8 + [C.POINTER(FrameOutputs)]
        L.oracle_render_frame_feedback.argtypes = ([C.c_void_p, C.POINTER(Camera), C.POINTER(Camera), C.c_void_p] + [C.c_uint32] * 6 +
                                                   [C.c_void_p, C.c_void_p, C.POINTER(FrameOutputs)])
        L.oracle_render_accumulate.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_void_p] + [C.c_uint32] * 7 + [C.c_void_p, C.c_void_p]
        L.oracle_post_create.restype = C.c_void_p
        L.oracle_post_create.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_post_destroy.argtypes = [C.c_void_p]
        L.oracle_post_set_threads.argtypes = [C.c_int]
        L.oracle_post_set_threads.restype = None
        L.oracle_post_pass.argtypes = [C.c_int, C.POINTER(PostSettings), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Camera), C.POINTER(Camera)] + [C.c_void_p] * 7
        L.oracle_post_frame.argtypes = [C.c_void_p, C.POINTER(PostSettings), C.c_uint32, C.POINTER(Camera), C.POINTER(Camera)] + [C.c_void_p] * 5
        L.oracle_wang_hash.restype = C.c_uint32
        L.oracle_wang_hash.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_pow22.restype = C.c_float
        L.oracle_pow22.argtypes = [C.c_float]
        L.oracle_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_halton23.argtypes = [C.c_uint32, C.c_void_p]
        L.oracle_bluenoise4x4.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_directional_light.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_primary_ray.argtypes = [C.POINTER(Camera)] + [C.c_uint32] * 5 + [C.c_void_p, C.c_void_p]
        L.oracle_map_to_hemisphere.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_oct_encode.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_intersect_triangle.restype = C.c_int
        L.oracle_intersect_triangle.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_sample_texture.argtypes = [C.POINTER(Texture), C.c_float, C.c_float, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def make_camera(position, forward, right, up, sensor_x=0.036, sensor_y=0.024, focal_length=0.016):
    cam = Camera()
    cam.position[:] = position
    cam.forward[:] = forward
    cam.right[:] = right
    cam.up[:] = up
    cam.sensor_size[0] = sensor_x
    cam.sensor_size[1] = sensor_y
    cam.focal_length = focal_length
    return cam


class Scene:
    """Owns an oracle scene built from GeometryStorage-layout numpy arrays (asset_load_system.h:16-39)."""

    def __init__(self, positions, normals, texcoords, indices, meshes, textures=(), materials=None):
        self._keep = [np.ascontiguousarray(positions, np.float32), np.ascontiguousarray(normals, np.float32),
                      np.ascontiguousarray(texcoords, np.float32), np.ascontiguousarray(indices, np.uint32),
                      np.ascontiguousarray(meshes, np.uint32).reshape(-1, 8)]
        d = SceneDesc()
        d.positions, d.normals, d.texcoords, d.indices, d.meshes = [_p(a) for a in self._keep]
        d.mesh_count = self._keep[4].shape[0]
        d.vertex_count = self._keep[0].size // 3
        d.index_count = self._keep[3].size
        texs = (Texture * max(1, len(textures)))()
        for i, t in enumerate(textures):
            t = np.ascontiguousarray(t, np.uint8)
            self._keep.append(t)
            texs[i].rgba8, texs[i].height, texs[i].width = _p(t), t.shape[0], t.shape[1]
        d.textures, d.texture_count = C.cast(texs, C.c_void_p), len(textures)
        if materials is not None:
            m = np.ascontiguousarray(materials, np.float32).reshape(-1, 12)
            self._keep.append(m)
            d.materials = _p(m)
        self.h = lib().oracle_scene_create(C.byref(d))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_scene_destroy(self.h)
            self.h = None

    def render_frame(self, cam, bluenoise, width, height, frame_count, num_bounces, flags=0, threads=1, feedback=None, rows=None):
        """feedback = (prev_cam, prev_normal_depth, color_history) turns the G-buffer feedback branch on (rt_indirect.hlsl:116-145).
        rows = (y0, y1): only those rows are rendered (the rest of every plane stays zero; rays counts the rendered rows)."""
        names = ("gbuffer_geo", "direct", "albedo", "normal_depth", "indirect", "combined")
        bufs = {n: np.zeros((height, width, 4), np.float32) for n in names}
        out = FrameOutputs()
        for n in names:
            setattr(out, n, _p(bufs[n]))
        if flags & FLAG_LOWRES_INDIRECT:
            bufs["indirect_lowres"] = np.zeros((height // 2, width // 2, 4), np.float32)
            out.indirect_lowres = _p(bufs["indirect_lowres"])
        bn = np.ascontiguousarray(bluenoise, np.uint8)
        if feedback is not None:
            prev_cam, pnd, hist = feedback
            pnd, hist = np.ascontiguousarray(pnd, np.float32), np.ascontiguousarray(hist, np.float32)
            assert pnd.shape == hist.shape == (height, width, 4)
            rc = lib().oracle_render_frame_feedback(self.h, C.byref(cam), C.byref(prev_cam), _p(bn), width, height, frame_count, num_bounces,
                                                    flags, threads, _p(pnd), _p(hist), C.byref(out))
        elif rows is not None:
            rc = lib().oracle_render_frame_rows(self.h, C.byref(cam), _p(bn), width, height, frame_count, num_bounces, flags, threads,
                                                int(rows[0]), int(rows[1]), C.byref(out))
        else:
            rc = lib().oracle_render_frame(self.h, C.byref(cam), _p(bn), width, height, frame_count, num_bounces, flags, threads, C.byref(out))
        if rc:
            raise RuntimeError("oracle_render_frame rc=%d" % rc)
        bufs["rays"] = tuple(int(x) for x in out.rays)
        return bufs

    def render_accumulate(self, cam, bluenoise, width, height, frame_begin, n_frames, num_bounces, flags=0, threads=1):
        acc = np.zeros((height, width, 4), np.float32)
        rays = (C.c_uint64 * 3)()
        bn = np.ascontiguousarray(bluenoise, np.uint8)
        rc = lib().oracle_render_accumulate(self.h, C.byref(cam), _p(bn), width, height, frame_begin, n_frames, num_bounces, flags,
                                            threads, _p(acc), C.cast(rays, C.c_void_p))
        if rc:
            raise RuntimeError("oracle_render_accumulate rc=%d" % rc)
        return acc, tuple(int(x) for x in rays)


class PostChain:
    """History-carrying reconstruction chain (Gather .. TAA) of the oracle."""

    def __init__(self, width, height, threads=1):
        self.w, self.h = width, height
        self.threads = threads
        self.handle = lib().oracle_post_create(width, height)

    def __del__(self):
        if getattr(self, "handle", None):
            lib().oracle_post_destroy(self.handle)
            self.handle = None

    def frame(self, settings, frame_count, cam, prev_cam, planes):
        a = [np.ascontiguousarray(planes[k], np.float32)
             for k in ("indirect_lowres" if settings.lowres_indirect else "indirect", "direct", "albedo", "normal_depth")]
        out = np.zeros((self.h, self.w, 4), np.float32)
        lib().oracle_post_set_threads(self.threads)
        rc = lib().oracle_post_frame(self.handle, C.byref(settings), frame_count, C.byref(cam), C.byref(prev_cam), _p(a[0]), _p(a[1]),
                                     _p(a[2]), _p(a[3]), _p(out))
        if rc:
            raise RuntimeError("oracle_post_frame rc=%d" % rc)
        return out


# ---- pure functions ----
def post_pass(kind, settings, inputs, arg=0, cam=None, prev_cam=None):
    """One pass of the reconstruction chain on full-resolution [h, w, 4] images: kind 0 Gather, 1 Accumulate (arg = frame_count),
    2 BlurDisocclusion, 3 Blur (arg = stride), 4 TAA.  Returns (out0, out1)."""
    ins = [np.ascontiguousarray(a, np.float32) if a is not None else None for a in inputs] + [None] * (5 - len(inputs))
    h, w = ins[0].shape[:2]
    out0, out1 = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    rc = lib().oracle_post_pass(kind, C.byref(settings), w, h, arg, C.byref(cam) if cam is not None else None,
                                C.byref(prev_cam) if prev_cam is not None else None, *[(_p(a) if a is not None else None) for a in ins],
                                _p(out0), _p(out1))
    if rc:
        raise RuntimeError("oracle_post_pass rc=%d" % rc)
    return out0, out1


def halton23(fc):
    o = np.zeros(2, np.float32); lib().oracle_halton23(fc, _p(o)); return o


def wang_hash(x, y):
    return int(lib().oracle_wang_hash(x, y))


def bluenoise4x4(bn, x, y, count):
    o = np.zeros(2, np.float32); bn = np.ascontiguousarray(bn, np.uint8); lib().oracle_bluenoise4x4(_p(bn), x, y, count, _p(o)); return o


def directional_light(count):
    d = np.zeros(3, np.float32); i = np.zeros(3, np.float32); lib().oracle_directional_light(count, _p(d), _p(i)); return d, i


def primary_ray(cam, x, y, w, h, fc):
    o = np.zeros(3, np.float32); d = np.zeros(3, np.float32); lib().oracle_primary_ray(C.byref(cam), x, y, w, h, fc, _p(o), _p(d)); return o, d


def map_to_hemisphere(s, n):
    s = np.asarray(s, np.float32); n = np.asarray(n, np.float32); o = np.zeros(3, np.float32)
    lib().oracle_map_to_hemisphere(_p(s), _p(n), _p(o)); return o


def sincos(x):
    s = C.c_float(); c = C.c_float(); lib().oracle_sincos(x, C.byref(s), C.byref(c)); return s.value, c.value


def pow22(x):
    return float(lib().oracle_pow22(x))


def oct_encode(n):
    n = np.asarray(n, np.float32); o = np.zeros(2, np.float32); lib().oracle_oct_encode(_p(n), _p(o)); return o


def intersect_triangle(o, d, tmin, tmax, v0, v1, v2):
    a = [np.asarray(x, np.float32) for x in (o, d, v0, v1, v2)]
    t = C.c_float(); u = C.c_float(); v = C.c_float()
    hit = lib().oracle_intersect_triangle(_p(a[0]), _p(a[1]), tmin, tmax, _p(a[2]), _p(a[3]), _p(a[4]), C.byref(t), C.byref(u), C.byref(v))
    return (t.value, u.value, v.value) if hit else None


def sample_texture(tex, u, v):
    tex = np.ascontiguousarray(tex, np.uint8); t = Texture(_p(tex), tex.shape[1], tex.shape[0]); o = np.zeros(3, np.float32)
    lib().oracle_sample_texture(C.byref(t), u, v, _p(o)); return o
