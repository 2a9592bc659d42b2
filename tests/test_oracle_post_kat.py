"""Known answers for the passes of the reconstruction chain, evaluated independently of the oracle: every function below is a
float64 numpy transcription of the reference's shader text, statement by statement, with the file and lines it follows
(/root/reference/src/core/shaders/...), written against the HLSL only -- not against oracle/cap_oracle_post.cpp.  The oracle's
single-pass entry (oracle_post_pass) must reproduce them to fp32 accuracy on random inputs, pixel by pixel, for interior, edge and
background pixels.  This pins what a shared misreading of the HLSL could otherwise hide (tap order and bounds, which sigma
multiplies what, the firefly clamp, the variance formulas, the history blend, the clip box), one pass at a time.  No GPU."""
import math

import numpy as np
import pytest

from oracle import cap_oracle as O

EPS = 1e-8  # math_functions.h:4


# ---------------------------------------------------------------- shader text, float64
def luminance(c):  # math_functions.h luminance == color_space.h Luminance
    return 0.299 * c[0] + 0.587 * c[1] + 0.114 * c[2]


def oct_decode(f):  # math_functions.h OctDecode
    fx, fy = f[0] * 2.0 - 1.0, f[1] * 2.0 - 1.0
    n = np.array([fx, fy, 1.0 - abs(fx) - abs(fy)])
    t = min(max(-n[2], 0.0), 1.0)
    n[0] += -t if n[0] >= 0.0 else t
    n[1] += -t if n[1] >= 0.0 else t
    return n / np.linalg.norm(n)


def w_normal(n0, n1, s):  # eaw_edge_stopping.h
    return max(float(np.dot(n0, n1)), 0.0) ** s


def w_depth(dc, dp, s):
    t = 0.0 if s == 0.0 else abs(dc - dp) / s
    return math.exp(-t)


def w_luma(lc, lp, s):
    return math.exp(-abs(lc - lp) / s)


def load(img, x, y):  # RWTexture2D load: out of bounds reads return 0
    h, w = img.shape[:2]
    return img[y, x].astype(np.float64) if (0 <= x < w and 0 <= y < h) else np.zeros(4)


def hlsl_gather(s, color, nd, x, y):  # spatial_gather.hlsl:28-109, UPSCALE2X off
    h, w = color.shape[:2]
    cg = load(nd, x, y)
    cn, cd = oct_decode(cg[:2]), cg[3]
    cc = load(color, x, y)[:3]
    if cd < 1e-5:
        return np.append(cc, 0.0)
    s_depth, s_normal, s_luma = cd * s.gather_depth_sigma, s.gather_normal_sigma, s.gather_luma_sigma
    fc, tw = np.zeros(3), 0.0
    for dy in range(-3, 4):
        for dx in range(-3, 4):
            sx, sy = x + dx, y + dy
            if sx < 0 or sy < 0 or sx >= w or sy >= h:
                continue
            c = load(color, sx, sy)[:3]
            g = load(nd, sx, sy)
            if g[3] < 1e-5:
                continue
            wgt = w_normal(cn, oct_decode(g[:2]), s_normal) * w_depth(cd, g[3], s_depth * math.hypot(dx, dy)) * \
                w_luma(luminance(cc), luminance(c), s_luma)
            fc += wgt * c
            tw += wgt
    return np.append(cc if tw < EPS else fc / tw, 1.0)


def hlsl_blur(s, stride, color, nd, x, y):  # eaw_blur.hlsl:48-137; s.use_variance = the USE_VARIANCE define (:68, :114, :127)
    h, w = color.shape[:2]
    cg = load(nd, x, y)
    cn, cd = oct_decode(cg[:2]), cg[3]
    cv = load(color, x, y)
    cc, cvar = np.minimum(cv[:3], 10.0), (cv[3] if s.use_variance else 0.0)
    if cd < 1e-5:
        return np.append(cc, cvar)
    kw = (1.0, 2.0 / 3.0, 1.0 / 6.0)
    s_depth = cd * stride * s.eaw_depth_sigma
    s_normal = s.eaw_normal_sigma
    s_luma = s.eaw_luma_sigma * math.sqrt(max(0.0, cvar + EPS))
    fc, fv, tw = np.zeros(3), 0.0, 0.0
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            sx, sy = x + dx * stride, y + dy * stride
            if sx < 0 or sy < 0 or sx >= w or sy >= h:
                continue
            v = load(color, sx, sy)
            c = np.minimum(v[:3], 10.0)
            g = load(nd, sx, sy)
            if g[3] < 1e-5:
                continue
            lw, hw = 1.0, 1.0
            if s.use_variance:
                lw = w_luma(luminance(cc), luminance(c), s_luma)
                hw = kw[abs(dx)] * kw[abs(dy)]
            wgt = w_normal(cn, oct_decode(g[:2]), s_normal) * w_depth(cd, g[3], s_depth * math.hypot(dx, dy))
            fc += wgt * hw * lw * c
            tw += wgt * hw * lw
            if s.use_variance:
                fv += hw * hw * wgt * wgt * lw * lw * v[3]
    if tw < EPS:
        return np.append(cc, cvar)
    return np.append(fc / tw, fv / (tw * tw))


def hlsl_blur_disocclusion(s, color, nd, moments, x, y):  # eaw_blur.hlsl:142-223
    h, w = color.shape[:2]
    hist = load(moments, x, y)[3]
    cg = load(nd, x, y)
    cn, cd = oct_decode(cg[:2]), cg[3]
    cv = load(color, x, y)
    cc, cvar = np.minimum(cv[:3], 10.0), (cv[3] if s.use_variance else 0.0)  # :160-165
    if cd < 1e-5 or hist >= 8:
        return np.append(cc, cvar)
    s_depth, s_normal, s_luma = cd * s.eaw_depth_sigma, s.eaw_normal_sigma, s.eaw_luma_sigma
    fc, fm, tw = np.zeros(3), np.zeros(2), 0.0
    for dy in range(-3, 4):
        for dx in range(-3, 4):
            sx, sy = x + dx, y + dy
            if sx < 0 or sy < 0 or sx >= w or sy >= h:
                continue
            c = np.minimum(load(color, sx, sy)[:3], 10.0)
            g = load(nd, sx, sy)
            m = load(moments, sx, sy)[:2]
            if g[3] < 1e-5:
                continue
            wgt = w_normal(cn, oct_decode(g[:2]), s_normal) * w_depth(cd, g[3], s_depth * math.hypot(dx, dy)) * \
                w_luma(luminance(cc), luminance(c), s_luma)
            fc += wgt * c
            fm += wgt * m
            tw += wgt
    col = cc if tw < EPS else fc / tw
    cm = np.zeros(2) if tw < EPS else fm / tw
    return np.append(col, (8.0 / hist) * abs(cm[1] - cm[0] * cm[0]))


def uv_to_xy(uv, w, h):  # utils.h:6-10
    return np.minimum(np.array([uv[0] * w, uv[1] * h]), np.array([w - 1.0, h - 1.0]))


def xy_to_uv(xy, w, h):  # utils.h:13-16
    return np.clip(np.array([xy[0] / w, xy[1] / h]), 0.0, 1.0)


def sample_bilinear(img, uv):  # utils.h:20-35; uint(floor(xy)) of a negative value taken as 0 (stated choice)
    h, w = img.shape[:2]
    xy = uv_to_xy(uv, w, h) - 0.5
    ux, uy = max(int(math.floor(xy[0])), 0), max(int(math.floor(xy[1])), 0)
    wx, wy = xy[0] - math.floor(xy[0]), xy[1] - math.floor(xy[1])
    v00, v01, v10, v11 = load(img, ux, uy)[:3], load(img, ux, uy + 1)[:3], load(img, ux + 1, uy)[:3], load(img, ux + 1, uy + 1)[:3]
    a, b = v00 + wx * (v10 - v00), v01 + wx * (v11 - v01)
    return a + wy * (b - a)


def cubic(x, b, c):  # math_functions.h cubic
    x2, x3, y = x * x, x * x * x, 0.0
    if x < 1.0:
        y = (12.0 - 9.0 * b - 6.0 * c) * x3 + (-18.0 + 12.0 * b + 6.0 * c) * x2 + (6.0 - 2.0 * b)
    elif x <= 2.0:
        y = (-b - 6.0 * c) * x3 + (6.0 * b + 30.0 * c) * x2 + (-12.0 * b - 48.0 * c) * x + (8.0 * b + 24.0 * c)
    return y / 6.0


def resample_bicubic(img, uv):  # temporal_accumulation.hlsl:39-66
    h, w = img.shape[:2]
    c = uv_to_xy(uv, w, h)
    filt, tw = np.zeros(3), 0.0
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            cur = c + np.array([i, j], np.float64)
            if cur[0] < 0 or cur[1] < 0 or cur[0] >= w or cur[1] >= h:
                continue
            v = sample_bilinear(img, xy_to_uv(cur, w, h))
            d = np.abs(cur - c)
            wt = cubic(d[0], 0, 0.5) * cubic(d[1], 0, 0.5) / (1.0 + luminance(v))
            filt += wt * v
            tw += wt
    return filt / tw if tw > 1e-5 else np.zeros(3)


def cam_vec(cam, name):
    return np.array(list(getattr(cam, name)), np.float64)


def reconstruct_world_position(cam, uv, depth):  # camera.h:64-80
    cs = (uv - 0.5) * np.array([cam.sensor_size[0], cam.sensor_size[1]], np.float64)
    d = cam.focal_length * cam_vec(cam, "forward") + cs[0] * cam_vec(cam, "right") + cs[1] * cam_vec(cam, "up")
    return cam_vec(cam, "position") + depth * d / np.linalg.norm(d)


def image_plane_uv(cam, pos):  # camera.h:8-37
    o = cam_vec(cam, "position")
    d = (pos - o) / np.linalg.norm(pos - o)
    n = cam_vec(cam, "forward") / np.linalg.norm(cam_vec(cam, "forward"))
    p = o + n * cam.focal_length
    t = np.dot(n, p - o) / np.dot(n, d)
    ipd = o + t * d - p
    u = np.dot(cam_vec(cam, "right"), ipd) / (0.5 * cam.sensor_size[0])
    v = np.dot(cam_vec(cam, "up"), ipd) / (0.5 * cam.sensor_size[1])
    return 0.5 * np.array([u, v]) + 0.5


def closest_depth(g, xy):  # temporal_accumulation.hlsl:179-205
    h, w = g.shape[:2]
    closest = load(g, int(xy[0]), int(xy[1]))[3]
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            tx, ty = int(xy[0]) + dx, int(xy[1]) + dy
            if tx >= w or ty >= h or tx < 0 or ty < 0:
                continue
            v = load(g, tx, ty)[3]
            if v != 0.0 and v < closest:
                closest = v
    return closest


def hlsl_accumulate(s, frame_count, cam, prev_cam, color, nd, chist, mhist, prev_nd, x, y):  # temporal_accumulation.hlsl:213-325
    h, w = nd.shape[:2]
    uv = (np.array([x, y], np.float64) + 0.5) / np.array([w, h], np.float64)
    g = load(nd, x, y)

    def fresh():
        c = sample_bilinear(color, uv)
        l = luminance(c)
        return np.append(c, 0.0), np.array([l, l * l, 0.0, 1.0])
    if g[3] < 1e-5:
        return fresh()
    hit = reconstruct_world_position(cam, uv, g[3])
    puv = image_plane_uv(prev_cam, hit)
    if puv[0] < 0 or puv[1] < 0 or puv[0] > 1 or puv[1] > 1 or frame_count == 0:
        return fresh()
    pxy = uv_to_xy(puv, w, h)
    cur_depth = np.linalg.norm(hit - cam_vec(prev_cam, "position"))
    if abs(closest_depth(prev_nd, pxy) - cur_depth) / cur_depth > 0.05:
        return fresh()
    alpha = s.temporal_upscale_feedback
    history = resample_bicubic(chist, puv)
    c = sample_bilinear(color, uv)
    hist_len = int(load(mhist, int(math.floor(pxy[0])), int(math.floor(pxy[1])))[3])
    if hist_len < 256:
        alpha = min(alpha, 1.0 - 1.0 / (hist_len + 1))
    mh = resample_bicubic(mhist, puv)
    l = luminance(c)
    m0, m1 = l + alpha * (mh[0] - l), l * l + alpha * (mh[1] - l * l)
    return np.append(c + alpha * (history - c), abs(m1 - m0 * m0)), np.array([m0, m1, 0.0, hist_len + 1.0])


def rgb2ycocg(c):  # color_space.h
    return np.array([c[0] / 4 + c[1] / 2 + c[2] / 4, c[0] / 2 - c[2] / 2, -c[0] / 4 + c[1] / 2 - c[2] / 4])


def ycocg2rgb(c):
    return np.clip(np.array([c[0] + c[1] - c[2], c[0] + c[2], c[0] - c[1] - c[2]]), 0.0, 1.0)


def tonemap(v):
    return v / (1.0 + luminance(v))


def inv_tonemap(v):
    return v / (1.0 - luminance(v))


def hlsl_taa(s, cam, prev_cam, color, nd, hist, x, y):  # temporal_accumulation.hlsl:362-447
    h, w = color.shape[:2]
    uv = (np.array([x, y], np.float64) + 0.5) / np.array([w, h], np.float64)
    g = load(nd, x, y)
    if g[3] < 1e-5:
        return np.append(sample_bilinear(color, uv), 1.0)
    hit = reconstruct_world_position(cam, uv, g[3])
    puv = image_plane_uv(prev_cam, hit)
    velocity = np.linalg.norm((puv - uv) * np.array([w, h], np.float64))
    if puv[0] < 0 or puv[1] < 0 or puv[0] > 1 or puv[1] > 1:
        return np.append(sample_bilinear(color, uv), 1.0)
    is_static = velocity < 1e-3
    alpha, scale = (0.98, 5.0) if is_static else (0.6, 0.75)
    alpha = min(s.taa_feedback, alpha)
    history = rgb2ycocg(tonemap(resample_bicubic(hist, puv)))
    c = rgb2ycocg(tonemap(sample_bilinear(color, uv)))
    center = rgb2ycocg(tonemap(sample_bilinear(color, xy_to_uv(np.array([x, y], np.float64), w, h))))  # :103
    m1, m2 = np.zeros(3), np.zeros(3)
    for i in range(-2, 3):
        for j in range(-2, 3):
            sx, sy = min(max(x + i, 0), w - 1), min(max(y + j, 0), h - 1)
            v = rgb2ycocg(tonemap(sample_bilinear(color, xy_to_uv(np.array([sx, sy], np.float64), w, h))))
            m1 += v
            m2 += v * v
    m1, m2 = m1 / 25.0, m2 / 25.0
    dev = np.sqrt(np.abs(m2 - m1 * m1)) * scale
    pmin, pmax = np.minimum(m1 - dev, center), np.maximum(m1 + dev, center)
    cc, radius = 0.5 * (pmin + pmax), 0.5 * (pmax - pmin)  # aabb.h:24-34
    dc = history - cc
    me = np.max(np.abs(dc / (radius + 1e-5)))
    if me > 1.0:
        history = cc + dc / me
    return np.append(inv_tonemap(ycocg2rgb(c + alpha * (history - c))), 1.0)


# ---------------------------------------------------------------- inputs
W, H = 24, 17


def scene(seed):
    rs = np.random.RandomState(seed)
    color = rs.uniform(0.0, 2.0, (H, W, 4)).astype(np.float32)
    color[3, 5, :3] = 30.0  # a firefly above the clamp of eaw_blur.hlsl:31
    color[..., 3] = rs.uniform(0.0, 0.3, (H, W))  # variance channel
    n = rs.normal(size=(H, W, 3)) * 0.15 + np.array([0.1, 0.2, 1.0])
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    nd = np.zeros((H, W, 4), np.float32)
    for y in range(H):
        for x in range(W):
            v = n[y, x] / np.abs(n[y, x]).sum()
            nd[y, x, 0:2] = v[0:2] * 0.5 + 0.5  # OctEncode of a normal with n.z >= 0
    nd[..., 3] = (3.0 + 0.05 * np.arange(W)[None, :] + 0.1 * rs.uniform(size=(H, W))).astype(np.float32)
    nd[2:4, 10:13, 3] = 0.0  # background pixels inside the image
    moments = rs.uniform(0.0, 1.0, (H, W, 4)).astype(np.float32)
    moments[..., 3] = rs.randint(1, 12, (H, W))  # history length: both sides of the threshold 8
    return color, nd, moments


PIXELS = [(7, 8), (0, 0), (W - 1, H - 1), (11, 3), (5, 3), (2, 14), (20, 6)]  # interior, corners, background, firefly, edges


def close(got, want, what):
    assert np.allclose(got, want, rtol=3e-4, atol=3e-6), "%s: oracle %s, shader text %s" % (what, got, want)


def camera(dx=0.0):
    return O.make_camera((0.1 + dx, 0.2, 0.3), (0.05, -0.02, -1.0), (-1.0, 0.0, -0.05), (0.0, 1.0, -0.02), 0.036, 0.036 * H / W, 0.03)


# ---------------------------------------------------------------- tests
def test_gather():
    color, nd, _ = scene(1)
    s = O.PostSettings()
    out, _ = O.post_pass(0, s, [color, nd])
    for x, y in PIXELS:
        close(out[y, x], hlsl_gather(s, color, nd, x, y), "Gather (%d, %d)" % (x, y))


@pytest.mark.parametrize("use_variance", [1, 0])
@pytest.mark.parametrize("stride", [1, 3, 7])
def test_blur(stride, use_variance):
    color, nd, _ = scene(2)
    s = O.PostSettings(eaw_luma_sigma=2.0, eaw_depth_sigma=1.5, eaw_normal_sigma=32.0, use_variance=use_variance)
    out, _ = O.post_pass(3, s, [color, nd], arg=stride)
    for x, y in PIXELS:
        close(out[y, x], hlsl_blur(s, stride, color, nd, x, y), "Blur stride %d (%d, %d)" % (stride, x, y))


@pytest.mark.parametrize("use_variance", [1, 0])
def test_blur_disocclusion(use_variance):
    color, nd, moments = scene(3)
    s = O.PostSettings(use_variance=use_variance)
    out, _ = O.post_pass(2, s, [color, nd, moments])
    for x, y in PIXELS:
        close(out[y, x], hlsl_blur_disocclusion(s, color, nd, moments, x, y), "BlurDisocclusion (%d, %d)" % (x, y))


@pytest.mark.parametrize("dx,frame", [(0.0, 3), (0.004, 3), (0.0, 0)])
def test_accumulate(dx, frame):
    color, nd, moments = scene(4)
    chist = scene(5)[0]
    cam, prev = camera(dx), camera(0.0)
    # depths that reproject consistently: the previous G-buffer sees the same surface
    prev_nd = nd.copy()
    s = O.PostSettings()
    o0, o1 = O.post_pass(1, s, [color, nd, chist, moments, prev_nd], arg=frame, cam=cam, prev_cam=prev)
    blended = 0
    for x, y in PIXELS + [(9, 9), (15, 12)]:
        wc, wm = hlsl_accumulate(s, frame, cam, prev, color, nd, chist, moments, prev_nd, x, y)
        close(o0[y, x], wc, "Accumulate colour (%d, %d)" % (x, y))
        close(o1[y, x], wm, "Accumulate moments (%d, %d)" % (x, y))
        blended += wm[3] > 1.0
    assert (blended > 0) == (frame != 0)  # the history branch is exercised, except on the first frame


@pytest.mark.parametrize("dx", [0.0, 0.004])
def test_taa(dx):
    color, nd, _ = scene(6)
    hist = scene(7)[0]
    cam, prev = camera(dx), camera(0.0)
    s = O.PostSettings()
    out, _ = O.post_pass(4, s, [color, nd, hist], cam=cam, prev_cam=prev)
    for x, y in PIXELS + [(9, 9)]:
        close(out[y, x], hlsl_taa(s, cam, prev, color, nd, hist, x, y), "TAA (%d, %d)" % (x, y))
