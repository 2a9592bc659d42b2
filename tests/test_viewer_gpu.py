"""The C++ host layer above the C ABI: capsaicin_amd/csrc/capsaicin.{h,cpp} keeps the reference's nine entry
points (src/core/include/capsaicin.h:25-36) and system order (src/core/src/capsaicin.cpp:38-62); csrc/viewer_main.cpp is the
headless src/viewer/main.cpp:50-107.  The viewer runs as a child process, exactly as a user would start it, and its PPM is
compared with the oracle's frames pushed through the composite blit (gamma 1 / 2.2 and vertical flip, simple.hlsl:40-46):
 * accumulate mode: plain mean of N frames;
 * --realtime --move: the reference's own per-frame pipeline (ray passes with G-buffer feedback -> reconstruction chain), camera
   translated every frame;
 * --realtime --script: the same with a fly-camera script -- per-frame keyboard displacement and mouse yaw / pitch replayed through
   ProcessInput() and the host layer's InputSystem step (input_system.cpp:49-148).
The per-pass timing table must carry the reference's timestamp labels (gui_system.cpp:94-104; raytracing_system.cpp:1024, 1099,
1207 and the reconstruction passes)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from capsaicin_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIEWER = os.path.join(ROOT, "capsaicin_amd", "capsaicin_viewer")
LABELS = ("RaytracePrimaryVisibility", "RT Direct lighting", "RT Indirect diffuse", "Spatial gather", "Temporal upscale", "EAW",
          "Combine illumination", "TAA")


def read_ppm(path):
    d = open(path, "rb").read()
    parts = d.split(b"\n", 3)
    assert parts[0] == b"P6" and parts[2] == b"255"
    w, h = (int(x) for x in parts[1].split())
    return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3)


def blit(img):
    """CompositeSystem's fullscreen pass (simple.hlsl:40-46): pow(c, 1 / 2.2), v flipped; 8-bit UNORM round-to-nearest."""
    v = np.power(np.maximum(img[..., :3].astype(np.float32), np.float32(0)), np.float32(1.0 / 2.2), dtype=np.float32)
    q = np.minimum(np.float32(255), np.floor(v * np.float32(255) + np.float32(0.5))).astype(np.uint8)
    return q[::-1]


def run_viewer(args, tmp_path):
    out = str(tmp_path / "frame.ppm")
    env = dict(os.environ, CAPSAICIN_ASSETS=os.path.join(ROOT, "assets"))
    p = subprocess.run([VIEWER, "--scene", os.path.join(ROOT, "assets", "cornell_box.obj"), "--out", out] + [str(a) for a in args],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return read_ppm(out), p.stderr


def same_up_to_one_code(got, want, what):
    """libm's powf and numpy's may differ in the last ulp, which moves a value sitting on an 8-bit rounding boundary by one code."""
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1, "%s: max code difference %d" % (what, int(diff.max()))
    assert (diff != 0).mean() < 1e-3, "%s: %.4f %% of the values differ" % (what, 100.0 * float((diff != 0).mean()))


def oracle_scene(cornell_path):
    from oracle import cap_oracle as O
    from oracle import obj_oracle
    g = obj_oracle.load_geometry(cornell_path)
    return O, O.Scene(g["positions"], g["normals"], g["texcoords"], g["indices"], g["meshes"])


def ocam_of(O, cam):
    return O.make_camera(tuple(cam.position), tuple(cam.forward), tuple(cam.right), tuple(cam.up), cam.sensor_size[0], cam.sensor_size[1],
                         cam.focal_length)


def test_accumulate_mode(native_lib, bluenoise, cornell_path, tmp_path):
    assert os.path.exists(VIEWER), "capsaicin_viewer is built by capsaicin_amd/csrc/Makefile"
    O, sc = oracle_scene(cornell_path)
    w, h, n, D = 136, 80, 5, 2
    got, log = run_viewer(["--width", w, "--height", h, "--frames", n, "--bounces", D], tmp_path)
    cam = capi.cornell_camera(w, h)  # the viewer's built-in Cornell view is assets/scene_config.json's
    acc, rays = sc.render_accumulate(ocam_of(O, cam), bluenoise, w, h, 0, n, D, threads=8)
    mean = acc[..., :3] / np.float32(n)
    same_up_to_one_code(got, blit(mean), "accumulated frame")
    for label in LABELS:
        assert label + ":" in log, "TimingsReport lacks the reference's label %r" % label
    # the ray counters of the session, as printed
    assert "rays primary/extension/shadow = %d/%d/%d" % rays in log


@pytest.mark.parametrize("feedback", [True, False])
def test_realtime_pipeline_with_camera_motion(native_lib, bluenoise, cornell_path, tmp_path, feedback):
    O, sc = oracle_scene(cornell_path)
    w, h, n, D = 120, 88, 6, 2
    move = (0.02, 0.0, -0.03)
    args = ["--width", w, "--height", h, "--frames", n, "--bounces", D, "--realtime", "--move"] + list(move)
    got, log = run_viewer(args + ([] if feedback else ["--no-feedback"]), tmp_path)
    chain = O.PostChain(w, h)
    s = O.PostSettings()
    base = capi.cornell_camera(w, h)
    prev_nd = np.zeros((h, w, 4), np.float32)
    hist = np.zeros((h, w, 4), np.float32)
    prev = None
    want = None
    for f in range(n):
        cam = capi.CameraData.from_buffer_copy(bytes(base))
        for k in range(3):  # the viewer adds `move` to the float position after every Render()
            p = np.float32(base.position[k])
            for _ in range(f):
                p = np.float32(p + np.float32(move[k]))
            cam.position[k] = p
        prev = prev if prev is not None else cam  # the first frame has no predecessor (camera_system.cpp:104-118)
        fb = (ocam_of(O, prev), prev_nd, hist) if feedback else None
        ref = sc.render_frame(ocam_of(O, cam), bluenoise, w, h, f, D, threads=8, feedback=fb)
        want = chain.frame(s, f, ocam_of(O, cam), ocam_of(O, prev), ref)
        prev, prev_nd, hist = cam, ref["normal_depth"], want
    same_up_to_one_code(got, blit(want), "frame %d of the realtime loop" % (n - 1))
    for label in LABELS:
        assert label + ":" in log


def test_realtime_pipeline_with_fly_camera_script(native_lib, bluenoise, cornell_path, tmp_path):
    """--script: the replay of what InputSystem would have seen (input_system.cpp:49-148), with ROTATION: per frame the viewer hands
    a ScriptedInput to ProcessInput(), and the host layer's InputSystem step turns the accumulated (pitch, yaw) into the camera basis
    the way HandleMouse does -- forward = (0, 0, 1) * XMMatrixRotationRollPitchYaw(pitch, yaw, 0) = (cos p sin y, -sin p, cos p cos y),
    right = normalize(-cross(forward, (0, 1, 0))), up = cross(forward, right) (input_system.cpp:130-146) -- then moves along the new
    axes like HandleKeyboard.  Checked twice: the cameras the viewer reports against this derivation in float64 (to 1e-6: libm's
    sinf / cosf vs numpy's), and the final frame of the real-time loop (G-buffer feedback on, reconstruction chain) against the
    oracle fed with exactly the reported cameras."""
    O, sc = oracle_scene(cornell_path)
    w, h, n, D = 120, 88, 6, 2
    steps = [(0.0, 0.0, 0.0, 0.0, 0.0), (0.01, 0.0, 0.02, 1.5, 0.0), (0.0, 0.005, 0.02, 1.5, -0.75), (0.0, 0.0, 0.0, 0.0, 0.0),
             (-0.02, 0.0, 0.03, -2.0, 0.5), (0.0, 0.0, 0.01, 0.0, 0.0)]
    script = tmp_path / "fly.txt"
    script.write_text("# right up forward dyaw dpitch\n" + "".join("%g %g %g %g %g\n" % s_ for s_ in steps))
    got, log = run_viewer(["--width", w, "--height", h, "--frames", n, "--bounces", D, "--realtime", "--script", str(script), "--print-cameras"], tmp_path)
    reported = {}
    for line in log.splitlines():
        if line.startswith("camera "):
            head, vals = line.split(":")
            reported[int(head.split()[1])] = [float.fromhex(v) for v in vals.split()]
    assert sorted(reported) == list(range(n))
    # the derivation, in float64, from the Cornell view (forward (0, 0, -1): yaw 180 degrees, pitch 0)
    base = capi.cornell_camera(w, h)
    pos = np.float64(list(base.position))
    fwd, right, up = (np.float64(list(v)) for v in (base.forward, base.right, base.up))
    yaw, pitch = 180.0, 0.0
    for f, (mr, mu, mf, dyaw, dpitch) in enumerate(steps):
        if dyaw != 0.0 or dpitch != 0.0:
            yaw, pitch = yaw + dyaw, pitch + dpitch
            p_, y_ = np.radians(pitch), np.radians(yaw)
            fwd = np.float64([np.cos(p_) * np.sin(y_), -np.sin(p_), np.cos(p_) * np.cos(y_)])
            fwd /= np.linalg.norm(fwd)
            right = -np.cross(fwd, (0.0, 1.0, 0.0))
            right /= np.linalg.norm(right)
            up = np.cross(fwd, right)
        pos = pos + right * mr + fwd * mf + up * mu
        assert np.allclose(reported[f], np.concatenate([pos, fwd, right, up]), rtol=0, atol=2e-6), f
    assert abs(np.dot(fwd, np.float64(list(base.forward)))) < 0.9999  # the view did turn

    def cam_of(vals):
        c = capi.CameraData.from_buffer_copy(bytes(base))
        c.position[:], c.forward[:], c.right[:], c.up[:] = vals[0:3], vals[3:6], vals[6:9], vals[9:12]
        return c

    chain, s = O.PostChain(w, h), O.PostSettings()
    prev_nd, hist = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    prev = want = None
    for f in range(n):
        cam = cam_of(reported[f])
        prev = prev if prev is not None else cam
        ref = sc.render_frame(ocam_of(O, cam), bluenoise, w, h, f, D, threads=8, feedback=(ocam_of(O, prev), prev_nd, hist))
        want = chain.frame(s, f, ocam_of(O, cam), ocam_of(O, prev), ref)
        prev, prev_nd, hist = cam, ref["normal_depth"], want
    same_up_to_one_code(got, blit(want), "frame %d of the scripted fly-through" % (n - 1))


def test_png_tga_and_jpeg_textures_through_the_host_layer(native_lib, bluenoise, tmp_path):
    """AssetLoadSystem + TextureSystem of the host layer on a textured scene whose textures are PNG, TGA and JPEG files
    (texture_system.cpp:41-45 loads any stb format): decoded by the layer's own decoders and rendered like the oracle renders the
    same texels.  The JPEG is a committed fixture whose texels are what the reference's decoder made of it (tests/golden/images)."""
    PIL = pytest.importorskip("PIL.Image")
    from oracle import cap_oracle as O
    rs = np.random.RandomState(4)
    tex = [rs.randint(0, 256, (16, 16, 3)).astype(np.uint8), rs.randint(0, 256, (8, 32, 3)).astype(np.uint8)]
    (tmp_path / "textures").mkdir()
    PIL.fromarray(tex[0], "RGB").save(str(tmp_path / "textures" / "a.png"))
    PIL.fromarray(tex[1], "RGB").save(str(tmp_path / "textures" / "b.tga"), compression="tga_rle")
    golden = os.path.join(ROOT, "tests", "golden", "images")
    shutil.copy(os.path.join(golden, "rgb420_optimised.jpg"), tmp_path / "textures" / "c.jpg")
    (tmp_path / "s.mtl").write_text("newmtl ma\nmap_Kd a.png\nnewmtl mb\nmap_Kd b.tga\nnewmtl mc\nmap_Kd c.jpg\n")
    (tmp_path / "s.obj").write_text("\n".join([
        "mtllib s.mtl", "o floor", "v -2 0 -2", "v 2 0 -2", "v 2 0 2", "v -2 0 2", "vn 0 1 0", "vt 0 0", "vt 3 0", "vt 3 3", "vt 0 3",
        "usemtl ma", "f 1/1/1 4/4/1 3/3/1 2/2/1",
        "o wall", "v -2 0 -2", "v 2 0 -2", "v 2 3 -2", "v -2 3 -2", "vn 0 0 1", "usemtl mb", "f 5/1/2 6/2/2 7/3/2 8/4/2",
        "o side", "v -2 0 2", "v -2 0 -2", "v -2 3 -2", "v -2 3 2", "vn 1 0 0", "usemtl mc", "f 9/1/3 10/2/3 11/3/3 12/4/3", ""]))
    w, h, n, D = 128, 72, 3, 2
    view = (0.3, 1.2, 3.5, 0.0, -0.2, -1.0, 0.03)
    out = str(tmp_path / "t.ppm")
    env = dict(os.environ, CAPSAICIN_ASSETS=os.path.join(ROOT, "assets"))
    # CAPSAICIN_ASSETS also names where textures/ is looked up: point it at a directory holding both the blue noise and the textures
    shutil.copy(os.path.join(ROOT, "assets", "bluenoise256.rgba"), tmp_path / "bluenoise256.rgba")
    env["CAPSAICIN_ASSETS"] = str(tmp_path)
    p = subprocess.run([VIEWER, "--scene", str(tmp_path / "s.obj"), "--out", out, "--width", str(w), "--height", str(h), "--frames", str(n),
                        "--bounces", str(D), "--camera"] + [str(v) for v in view], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "missing" not in p.stderr, p.stderr[-2000:]
    geo = capi.Geometry(str(tmp_path / "s.obj"), str(tmp_path))
    assert geo.texture_names == ["a.png", "b.tga", "c.jpg"]
    rgba = [np.concatenate([t, np.full(t.shape[:2] + (1,), 255, np.uint8)], -1) for t in tex]
    rgba.append(np.load(os.path.join(golden, "expected.npz"))["rgb420_optimised.jpg"])
    sc = O.Scene(geo.positions, geo.normals, geo.texcoords, geo.indices, geo.meshes, textures=rgba)
    cam = capi.camera_from_config(dict(position=view[0:3], forward=view[3:6], focal_length=view[6], sensor_x=0.036), w, h)
    acc, _ = sc.render_accumulate(ocam_of(O, cam), bluenoise, w, h, 0, n, D, threads=8)
    same_up_to_one_code(read_ppm(out), blit(acc[..., :3] / np.float32(n)), "textured frame")
