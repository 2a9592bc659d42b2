#!/usr/bin/env python3
"""Procedural "Sponza-class" scene (BASELINE.json configs[3]; SURVEY.md 8d: the reference viewer loads ../../../assets/sponza.obj,
main.cpp:88, which is not in the repository): an atrium with a tessellated floor, walls with arcades, two storeys of columns,
arches, draped cloth and a few vases — about 260 k triangles in 12 meshes, every mesh with its own MTL material and a
procedural RGBA8 texture (binary PPM, decoded natively by the host library).  Deterministic: PCG32 streams seeded 0x5EED.

    python tools/make_sponza_class.py OUT_DIR [--scale 1.0]

writes OUT_DIR/sponza_class.obj, OUT_DIR/sponza_class.mtl and OUT_DIR/textures/*.ppm.  --scale < 1 shrinks the tessellation
(tests use small scales; scale 1.0 gives ~262 k triangles).
"""
import argparse
import os

import numpy as np


class PCG32:
    def __init__(self, seed=0x5EED, seq=1):
        self.state, self.inc = 0, (seq << 1) | 1
        self.next()
        self.state = (self.state + seed) & 0xFFFFFFFFFFFFFFFF
        self.next()

    def next(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & 0xFFFFFFFFFFFFFFFF
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform(self, n):
        return np.array([self.next() for _ in range(n)], np.float64) / 4294967296.0


def grid(nu, nv, f):
    """Tessellated parametric patch: f(u, v) -> (pos[...,3], normal[...,3]); returns verts, normals, uvs, quads->tris."""
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    p, n = f(u, v)
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    tris = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
    return p.reshape(-1, 3), n.reshape(-1, 3), np.stack([u, v], -1).reshape(-1, 2), tris


def merge(parts):
    P, N, T, F, off = [], [], [], [], 0
    for p, n, t, f in parts:
        P.append(p), N.append(n), T.append(t), F.append(f + off)
        off += len(p)
    return np.concatenate(P), np.concatenate(N), np.concatenate(T), np.concatenate(F)


def plane(origin, eu, ev, nu, nv, uv_scale=1.0):
    origin, eu, ev = (np.asarray(x, np.float64) for x in (origin, eu, ev))
    nrm = np.cross(eu, ev)
    nrm /= np.linalg.norm(nrm)

    def f(u, v):
        return origin + u[..., None] * eu + v[..., None] * ev, np.broadcast_to(nrm, u.shape + (3,))
    p, n, t, tr = grid(nu, nv, f)
    return p, n, t * uv_scale, tr


def cylinder(base, radius, height, nseg, nring, bulge=0.0):
    base = np.asarray(base, np.float64)

    def f(u, v):
        ang = 2 * np.pi * u
        r = radius * (1.0 + bulge * np.sin(np.pi * v) ** 2)
        p = np.stack([base[0] + r * np.cos(ang), base[1] + height * v, base[2] + r * np.sin(ang)], -1)
        n = np.stack([np.cos(ang), np.zeros_like(ang), np.sin(ang)], -1)
        return p, n
    return grid(nseg, nring, f)


def arch(center, radius, tube, nseg, nring, axis="x"):
    center = np.asarray(center, np.float64)

    def f(u, v):
        a, b = np.pi * u, 2 * np.pi * v
        rr = radius + tube * np.cos(b)
        x, y, z = rr * np.cos(a), rr * np.sin(a), tube * np.sin(b)
        nx, ny, nz = np.cos(b) * np.cos(a), np.cos(b) * np.sin(a), np.sin(b)
        if axis == "x":
            return center + np.stack([x, y, z], -1), np.stack([nx, ny, nz], -1)
        return center + np.stack([z, y, x], -1), np.stack([nz, ny, nx], -1)
    return grid(nseg, nring, f)


def cloth(origin, width, drop, nu, nv, waves, phase):
    origin = np.asarray(origin, np.float64)

    def f(u, v):
        z = 0.18 * np.sin(2 * np.pi * waves * u + phase) * (0.3 + v)
        p = np.stack([origin[0] + width * u, origin[1] - drop * v, origin[2] + z], -1)
        dz = 0.18 * 2 * np.pi * waves * np.cos(2 * np.pi * waves * u + phase) * (0.3 + v) / width
        n = np.stack([-dz, np.zeros_like(dz), np.ones_like(dz)], -1)
        return p, n / np.linalg.norm(n, axis=-1, keepdims=True)
    return grid(nu, nv, f)


def vase(base, nseg, nring):
    base = np.asarray(base, np.float64)

    def f(u, v):
        ang = 2 * np.pi * u
        r = 0.25 + 0.2 * np.sin(np.pi * (0.15 + 0.85 * v)) ** 2 - 0.1 * v
        dr = 0.2 * 2 * np.sin(np.pi * (0.15 + 0.85 * v)) * np.cos(np.pi * (0.15 + 0.85 * v)) * np.pi * 0.85 - 0.1
        p = np.stack([base[0] + r * np.cos(ang), base[1] + 0.9 * v, base[2] + r * np.sin(ang)], -1)
        n = np.stack([np.cos(ang) * 0.9, -dr, np.sin(ang) * 0.9], -1)
        return p, n / np.linalg.norm(n, axis=-1, keepdims=True)
    return grid(nseg, nring, f)


def texture(kind, rng, size):
    y, x = np.mgrid[0:size, 0:size]
    if kind == "checker":
        a = ((x // (size // 16) + y // (size // 16)) % 2).astype(np.float64)
        img = np.stack([0.35 + 0.5 * a, 0.33 + 0.45 * a, 0.3 + 0.4 * a], -1)
    elif kind == "bricks":
        row = y // (size // 32)
        xx = (x + (row % 2) * (size // 16)) % (size // 8)
        mortar = (xx < size // 128) | (y % (size // 32) < size // 128)
        img = np.where(mortar[..., None], 0.75, np.array([0.62, 0.32, 0.25]))
        img = img * (0.85 + 0.15 * rng.uniform(size // 8)[(x // 8) % (size // 8)][..., None])
    elif kind == "marble":
        n = rng.uniform(64 * 64).reshape(64, 64)
        n = np.kron(n, np.ones((size // 64, size // 64)))
        v = 0.5 + 0.5 * np.sin((x + y) * 16 * np.pi / size + 6 * n)
        img = np.stack([0.75 + 0.2 * v, 0.72 + 0.2 * v, 0.68 + 0.22 * v], -1)
    else:  # cloth stripes
        c = rng.uniform(3)
        s = ((x // (size // 24)) % 2).astype(np.float64)
        img = np.stack([0.2 + 0.7 * c[0] * s + 0.1, 0.15 + 0.7 * c[1] * (1 - s), 0.2 + 0.6 * c[2] * s], -1)
    return (np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8)


def build(scale):
    s = lambda n: max(2, int(round(n * scale * 0.873)))  # 0.873: scale 1.0 lands on ~262 k triangles
    L, W, H = 24.0, 10.0, 9.0
    meshes = []  # (name, material, texture kind, geometry)
    meshes.append(("floor", "m_floor", "checker", plane((-L / 2, 0, -W / 2), (L, 0, 0), (0, 0, W), s(160), s(64), 8.0)))
    meshes.append(("ceiling_beams", "m_beams", "bricks", merge([plane((-L / 2 + i * L / 8, H, -W / 2), (0.4, 0, 0), (0, 0, W), s(4), s(40), 2.0) for i in range(9)])))
    meshes.append(("wall_north", "m_wall_n", "bricks", plane((-L / 2, 0, -W / 2), (0, H, 0), (L, 0, 0), s(72), s(160), 6.0)))
    meshes.append(("wall_south", "m_wall_s", "bricks", plane((-L / 2, 0, W / 2), (L, 0, 0), (0, H, 0), s(160), s(72), 6.0)))
    meshes.append(("wall_east", "m_wall_e", "marble", plane((L / 2, 0, -W / 2), (0, H, 0), (0, 0, W), s(72), s(64), 3.0)))
    meshes.append(("wall_west", "m_wall_w", "marble", plane((-L / 2, 0, -W / 2), (0, 0, W), (0, H, 0), s(64), s(72), 3.0)))
    cols = []
    for i in range(8):
        x = -L / 2 + 1.5 + i * (L - 3) / 7
        for z in (-W / 2 + 1.6, W / 2 - 1.6):
            cols.append(cylinder((x, 0, z), 0.32, 4.0, s(40), s(72), 0.08))
    meshes.append(("columns_lower", "m_col_lo", "marble", merge(cols)))
    cols = []
    for i in range(8):
        x = -L / 2 + 1.5 + i * (L - 3) / 7
        for z in (-W / 2 + 1.6, W / 2 - 1.6):
            cols.append(cylinder((x, 4.6, z), 0.24, 3.4, s(32), s(56), 0.05))
    meshes.append(("columns_upper", "m_col_up", "marble", merge(cols)))
    arcs = []
    for i in range(7):
        x = -L / 2 + 1.5 + (i + 0.5) * (L - 3) / 7
        for z in (-W / 2 + 1.6, W / 2 - 1.6):
            arcs.append(arch((x, 4.0, z), (L - 3) / 14, 0.18, s(48), s(20)))
    meshes.append(("arches", "m_arch", "bricks", merge(arcs)))
    meshes.append(("gallery", "m_gallery", "checker", merge([plane((-L / 2, 4.45, z0), (L, 0, 0), (0, 0, 1.9), s(160), s(16), 10.0) for z0 in (-W / 2, W / 2 - 1.9)])))
    cl = [cloth((-L / 2 + 2.5 + i * 4.4, 8.4, (-1) ** i * 1.2), 3.2, 3.6, s(72), s(64), 2 + i % 3, 0.7 * i) for i in range(5)]
    meshes.append(("drapes", "m_drapes", "cloth", merge(cl)))
    meshes.append(("vases", "m_vases", "marble", merge([vase((-L / 2 + 3 + i * 3.6, 0, (-1) ** i * 0.8), s(48), s(40)) for i in range(6)])))
    return meshes


def write(out_dir, scale=1.0, tex_size=1024):
    os.makedirs(os.path.join(out_dir, "textures"), exist_ok=True)
    meshes = build(scale)
    rng = PCG32(0x5EED, 7)
    ntri = 0
    with open(os.path.join(out_dir, "sponza_class.obj"), "w") as f, open(os.path.join(out_dir, "sponza_class.mtl"), "w") as m:
        f.write("# procedural Sponza-class scene, tools/make_sponza_class.py --scale %g\nmtllib sponza_class.mtl\n" % scale)
        voff = 1
        for name, mat, kind, (p, n, t, tris) in meshes:
            tex = "%s.ppm" % mat
            img = texture(kind, rng, tex_size)
            with open(os.path.join(out_dir, "textures", tex), "wb") as tf:
                tf.write(b"P6\n%d %d\n255\n" % (tex_size, tex_size))
                tf.write(img.tobytes())
            m.write("newmtl %s\nKd 0.8 0.8 0.8\nKs 0.04 0.04 0.04\nNs 32\nmap_Kd %s\n" % (mat, tex))
            f.write("o %s\nusemtl %s\n" % (name, mat))
            p32, n32, t32 = p.astype(np.float32), n.astype(np.float32), t.astype(np.float32)
            f.write("".join("v %.9g %.9g %.9g\n" % tuple(r) for r in p32))
            f.write("".join("vt %.9g %.9g\n" % tuple(r) for r in t32))
            f.write("".join("vn %.9g %.9g %.9g\n" % tuple(r) for r in n32))
            g = tris + voff
            f.write("".join("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, b, b, b, c, c, c) for a, b, c in g))
            voff += len(p)
            ntri += len(tris)
    return ntri


def arrays(scale=1.0, tex_size=1024):
    """The same scene as GeometryStorage-layout arrays, without the OBJ text in between (what `write` + the OBJ loader produce for
    it, vertex for vertex: every `f` line of `write` uses one index for position / uv / normal, so the loader's pools are the
    meshes' own vertex arrays): (positions [V,3], normals [V,3], texcoords [V,2] float32, mesh-local indices [I] uint32,
    mesh descriptors [12,8] uint32 in MeshComponent layout, textures: 12 x [tex_size, tex_size, 4] uint8).  For scenes the OBJ
    round trip would make gigabytes of text of: scale 4 is 4.2 M triangles, scale 8 is 16.8 M."""
    meshes = build(scale)
    rng = PCG32(0x5EED, 7)
    P, N, T, I, D, texs = [], [], [], [], [], []
    voff = ioff = 0
    for m, (name, mat, kind, (p, n, t, tris)) in enumerate(meshes):
        img = texture(kind, rng, tex_size)
        texs.append(np.concatenate([img, np.full(img.shape[:2] + (1,), 255, np.uint8)], -1))
        # the loader numbers a mesh's vertices in the order the faces first use them (asset_load_system.cpp:97-141)
        flat = tris.reshape(-1)
        uniq, first = np.unique(flat, return_index=True)
        order = uniq[np.argsort(first, kind="stable")]
        remap = np.zeros(len(p), np.int64)
        remap[order] = np.arange(len(order))
        p, n, t, tris = p[order], n[order], t[order], remap[flat].reshape(-1, 3)
        P.append(p.astype(np.float32)), N.append(n.astype(np.float32)), T.append(t.astype(np.float32))
        I.append(tris.astype(np.uint32).reshape(-1))
        D.append([len(p), voff, 3 * len(tris), ioff, m, m, 0, 0])
        voff += len(p)
        ioff += 3 * len(tris)
    return (np.concatenate(P), np.concatenate(N), np.concatenate(T), np.concatenate(I), np.uint32(D), texs)


def camera():
    """Viewpoint used by tests and the extra bench line: inside the hall, looking down its length."""
    import json
    cfg = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "scene_config.json")))
    c = cfg["sponza_class"]["camera"]
    return dict(position=tuple(c["position"]), forward=tuple(c["forward"]), focal_length=c["focal_length"])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--tex-size", type=int, default=1024)
    a = ap.parse_args()
    print("triangles:", write(a.out_dir, a.scale, a.tex_size))
