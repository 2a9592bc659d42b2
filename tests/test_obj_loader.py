"""Native OBJ/MTL ingestion (cap_obj_load) against the pure-Python restatement of asset_load_system.cpp:43-255."""
import os

import numpy as np
import pytest

from capsaicin_amd import capi
from oracle import obj_oracle


def _same(native, ref):
    assert np.array_equal(native.positions.view(np.uint32), ref["positions"].view(np.uint32))
    assert np.array_equal(native.normals.view(np.uint32), ref["normals"].view(np.uint32))
    assert np.array_equal(native.texcoords.view(np.uint32), ref["texcoords"].view(np.uint32))
    assert np.array_equal(native.indices, ref["indices"])
    assert np.array_equal(native.meshes, ref["meshes"])


def test_cornell_box(native_lib, cornell_path):
    g = capi.Geometry(cornell_path)
    _same(g, obj_oracle.load_geometry(cornell_path))
    assert g.meshes.shape == (8, 8) and g.positions.size == 64 * 3 and g.indices.size == 96
    assert "cornellbox.mtl" in g.warning and g.material_count == 0  # missing MTL is a warning only


OBJ_MIXED = """# mixed records
mtllib a.mtl missing.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 0.5 1
vt 0 0
vt 1 0
vt 1 1
vn 0 0 1
o first
usemtl red
f 1/1/1 2/2/1 3/3/1 4/1/1
f 1 2 5
g second group
usemtl tex
f -1//1 -2//1 -3//1
f 1/1 2/2 3/3 4/1 5/2
o empty
o third
usemtl nothing
s off
f 3/3/1 2/2/1 1/1/1
"""
MTL_A = """newmtl red
Kd 1 0 0
Ks 0.5 0.5 0.5
Ns 98
newmtl tex
Kd 1 1 1
map_Kd checker.ppm
Ke 1 2 3
"""


def test_mixed_records(native_lib, tmp_path):
    (tmp_path / "m.obj").write_text(OBJ_MIXED)
    (tmp_path / "a.mtl").write_text(MTL_A)
    g = capi.Geometry(str(tmp_path / "m.obj"))
    ref = obj_oracle.load_geometry(str(tmp_path / "m.obj"))
    _same(g, ref)
    assert g.meshes.shape[0] == 3  # the empty `o` produces no shape
    assert list(g.meshes[:, 5]) == [0xFFFFFFFF, 0, 0xFFFFFFFF]  # texture of material_ids[0] only (asset_load_system.cpp:146-150)
    assert g.texture_names == ["checker.ppm"] and g.material_count == 2 and "missing.mtl" in g.warning
    # fan triangulation: quad -> 2, pentagon -> 3 triangles
    assert list(g.meshes[:, 2]) == [9, 12, 3]
    m = g.materials()
    np.testing.assert_allclose(m[0, :3], (1, 0, 0))
    np.testing.assert_allclose(m[1, 8:11], (1, 2, 3))
    # explicit mtl directory (the reference passes "../../../assets/", asset_load_system.cpp:55)
    os.makedirs(tmp_path / "mats")
    (tmp_path / "mats" / "a.mtl").write_text(MTL_A.replace("checker.ppm", "other.ppm"))
    g2 = capi.Geometry(str(tmp_path / "m.obj"), str(tmp_path / "mats"))
    assert g2.texture_names == ["other.ppm"]
    _same(g2, obj_oracle.load_geometry(str(tmp_path / "m.obj"), str(tmp_path / "mats")))


def test_missing_normals_and_uvs_become_zero(native_lib, tmp_path):
    (tmp_path / "t.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    g = capi.Geometry(str(tmp_path / "t.obj"))
    assert np.all(g.normals == 0) and np.all(g.texcoords == 0) and g.meshes.shape[0] == 1  # asset_load_system.cpp:124-140


def test_errors_are_reported_not_swallowed(native_lib, tmp_path):
    with pytest.raises(capi.CapError):
        capi.Geometry(str(tmp_path / "does_not_exist.obj"))  # asset_load_system.cpp:57-67 throws
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n")
    with pytest.raises(capi.CapError):
        capi.Geometry(str(tmp_path / "bad.obj"))
    (tmp_path / "bad2.obj").write_text("v 0 0\n")
    with pytest.raises(capi.CapError):
        capi.Geometry(str(tmp_path / "bad2.obj"))


def test_empty_file(native_lib, tmp_path):
    (tmp_path / "e.obj").write_text("# nothing\n")
    g = capi.Geometry(str(tmp_path / "e.obj"))
    assert g.meshes.shape[0] == 0 and g.indices.size == 0


def test_scene_arrays_equal_the_obj_round_trip(native_lib, tmp_path):
    """tools/make_sponza_class.py arrays(): the in-memory form the multi-million-triangle scenes use (no OBJ text in between) is
    the GeometryStorage the native loader makes of the OBJ the same tool writes, bit for bit, textures included."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_sponza_class as gen
    gen.write(str(tmp_path), 0.1, 64)
    geo = capi.Geometry(os.path.join(str(tmp_path), "sponza_class.obj"))
    pos, nrm, uv, idx, meshes, texs = gen.arrays(0.1, 64)
    assert np.array_equal(geo.positions.view(np.uint32), pos.reshape(-1).view(np.uint32))
    assert np.array_equal(geo.normals.view(np.uint32), nrm.reshape(-1).view(np.uint32))
    assert np.array_equal(geo.texcoords.view(np.uint32), uv.reshape(-1).view(np.uint32))
    assert np.array_equal(geo.indices, idx) and np.array_equal(geo.meshes, meshes)
    for name, t in zip(geo.texture_names, texs):
        assert np.array_equal(capi.image_decode(open(os.path.join(str(tmp_path), "textures", name), "rb").read(), name), t)


def test_number_parsing_is_strtod(native_lib, tmp_path):
    """The loader parses the hot records in place (obj_loader.cpp fast_float: the exact-integer-times-power-of-ten path for short
    decimals, strtod for everything else).  What it must return is (float)strtod(token) for every token: checked against Python's
    correctly rounded float() on hand-picked edge cases and 30 000 random decimals of 1..19 digits, exponents to +-30, with and
    without fraction, sign and exponent forms; plus the prefix rule (garbage behind a number is ignored, as strtod does)."""
    import random
    rnd = random.Random(7)
    toks = ["0", "-0", "+0.0", "1", "-1", ".5", "5.", "+.5e1", "1e22", "1e23", "1e-22", "1e-23", "123456789012345", "1234567890123456",
            "12345678901234567890", "0.000000000000000000001", "3.4028235e38", "3.4028236e38", "1e39", "1.17549435e-38", "1e-45", "7e-46",
            "4.9e-324", "0.1", "0.2", "0.30000000000000004", "16777217", "16777216.5", "1E+5", "1e+05", "9007199254740993", "0x10", "1.5e",
            "2.5E-", "000123.4500", "-.0e5"]
    for _ in range(30000):
        nd = rnd.randint(1, 19)
        digs = "".join(rnd.choice("0123456789") for _ in range(nd))
        if rnd.random() < 0.7:
            k = rnd.randint(0, nd)
            digs = digs[:k] + "." + digs[k:]
        if digs in (".",):
            digs = "0."
        t = rnd.choice(["", "-", "+"]) + digs
        if rnd.random() < 0.5:
            t += rnd.choice("eE") + rnd.choice(["", "-", "+"]) + str(rnd.randint(0, 30))
        toks.append(t)
    while len(toks) % 3:
        toks.append("0")

    def want(t):
        if t == "0x10":
            return 16.0      # strtod reads hexadecimal
        if t in ("1.5e", "2.5E-"):
            return float(t.rstrip("eE-"))  # the longest prefix that is a number
        return float(t)

    lines = ["v %s %s %s" % tuple(toks[i:i + 3]) for i in range(0, len(toks), 3)]
    n = len(lines)
    obj = tmp_path / "numbers.obj"
    obj.write_text("\n".join(lines) + "\n" + "\n".join("f %d %d %d" % (i + 1, (i + 1) % n + 1, (i + 2) % n + 1) for i in range(n)) + "\n")
    geo = capi.Geometry(str(obj))
    # de-duplicated per (position index): the faces use every vertex, first use in order 1, 2, 3, 2->..., so map back through the indices
    pos = geo.positions.reshape(-1, 3)
    idx = geo.indices
    first = {}
    for k, f in enumerate(range(n)):
        for c, v in enumerate((f, (f + 1) % n, (f + 2) % n)):
            first.setdefault(v, idx[3 * k + c])
    with np.errstate(over="ignore"):
        for v in range(n):
            got = pos[first[v]]
            exp = np.array([want(t) for t in toks[3 * v:3 * v + 3]], np.float64).astype(np.float32)
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (v, toks[3 * v:3 * v + 3], got, exp)
    # garbage behind a number inside a token is ignored (strtod's prefix rule), a token that starts with none is an error
    (tmp_path / "g.obj").write_text("v 1.0abc 2 3\nv 0 0 0\nv 1 1 1\nf 1 2 3\n")
    assert capi.Geometry(str(tmp_path / "g.obj")).positions[0] == 1.0
    (tmp_path / "b.obj").write_text("v abc 2 3\n")
    with pytest.raises(capi.CapError, match="malformed number"):
        capi.Geometry(str(tmp_path / "b.obj"))
    (tmp_path / "c.obj").write_text("v abc 2\n")
    with pytest.raises(capi.CapError, match="expected 3 coordinates"):
        capi.Geometry(str(tmp_path / "c.obj"))


def test_threaded_parse_is_the_sequential_parse(native_lib, tmp_path):
    """cap_obj_load parses the hot records of a file above 1 MB on several threads (obj_loader.cpp load_obj_parallel) and falls back to
    the sequential parser on any anomaly: for a valid file the same bytes as with one thread (and the generator's own arrays), for a
    file with a late error the sequential parser's message with its line number, whatever the thread count."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_sponza_class as gen
    gen.write(str(tmp_path), 0.35, 64)
    path = os.path.join(str(tmp_path), "sponza_class.obj")
    assert os.path.getsize(path) > (1 << 20)  # above the threshold: the threaded path is what threads = 0 takes
    lib = capi.lib()
    try:
        got = {}
        for threads in (1, 0, 2, 3, 7):
            lib.cap_obj_set_threads(threads)
            got[threads] = capi.Geometry(path)
        pos, nrm, uv, idx, meshes, _ = gen.arrays(0.35, 64)
        for threads, g in got.items():
            assert np.array_equal(g.positions.view(np.uint32), pos.reshape(-1).view(np.uint32)), threads
            assert np.array_equal(g.normals.view(np.uint32), nrm.reshape(-1).view(np.uint32)), threads
            assert np.array_equal(g.texcoords.view(np.uint32), uv.reshape(-1).view(np.uint32)), threads
            assert np.array_equal(g.indices, idx) and np.array_equal(g.meshes, meshes), threads
            assert g.texture_names == got[1].texture_names and g.warning == got[1].warning, threads
        # a face that refers to a vertex the file only defines LATER (valid for no thread count), and a malformed number, both far into the file
        text = open(path).read()
        lines = text.split("\n")
        first_f = next(i for i, l in enumerate(lines) if l.startswith("f "))
        bad_index = lines[:first_f] + ["f 1 2 99999999"] + lines[first_f:]
        late = len(lines) * 3 // 4
        bad_number = lines[:late] + ["v 1.0 abc 2.0"] + lines[late:]
        for name, ls in (("bad_index.obj", bad_index), ("bad_number.obj", bad_number)):
            (tmp_path / name).write_text("\n".join(ls))
            msgs = []
            for threads in (1, 0, 5):
                lib.cap_obj_set_threads(threads)
                with pytest.raises(capi.CapError) as e:
                    capi.Geometry(str(tmp_path / name))
                msgs.append(str(e.value))
            assert msgs[0] == msgs[1] == msgs[2] and name in msgs[0] and ":" in msgs[0], msgs
    finally:
        lib.cap_obj_set_threads(0)
