"""Pure-Python restatement of the reference's OBJ ingestion (ORACLE, test infrastructure only).

Follows /root/reference/src/core/src/systems/asset_load_system.cpp:43-160 (LoadObjFile: per-shape
(vertex, normal, texcoord)-tuple de-duplication, zeros for absent normals/uvs, texture index of
material_ids[0]) and :162-255 (CreateGeometryStorage: pooled buffers + 32-byte mesh descriptors).

tinyobjloader itself is an un-vendored, un-pinned submodule (.gitmodules:4-6, directory empty), so the
part of its published behaviour the call site relies on is restated here: `v/vn/vt` records, 1-based and
negative (relative) face indices, `f` polygons triangulated as a fan (i0, i[k-1], i[k]), a new shape at
every `o`/`g` record (empty shapes dropped), per-face material ids from `usemtl` resolved against the
materials of the `mtllib` files that could be opened (a missing file is only a warning and leaves every
id at -1), `map_Kd` -> diffuse_texname.  Triangulation order / shape splitting are "parity unpinned".
"""
import os
import numpy as np

INVALID = 0xFFFFFFFF


def parse_mtl(path):
    mats = []  # list of dict(name, diffuse_texname, Kd, Ks, Ns, Ke)
    cur = None
    with open(path) as f:
        for raw in f:
            t = raw.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "newmtl":
                cur = dict(name=" ".join(t[1:]), diffuse_texname="", Kd=(0.0, 0.0, 0.0), Ks=(0.0, 0.0, 0.0), Ns=1.0, Ke=(0.0, 0.0, 0.0))
                mats.append(cur)
            elif cur is not None and t[0] in ("Kd", "Ks", "Ke"):
                cur[t[0]] = tuple(float(x) for x in t[1:4])
            elif cur is not None and t[0] == "Ns":
                cur["Ns"] = float(t[1])
            elif cur is not None and t[0] == "map_Kd":
                cur["diffuse_texname"] = t[-1]
    return mats


def parse_obj(path, mtl_dir=None):
    """-> (attrib dict, shapes list, materials list, warn str).  shapes: dict(name, indices[(v,vn,vt)], material_ids)."""
    mtl_dir = os.path.dirname(path) if mtl_dir is None else mtl_dir
    V, VN, VT = [], [], []
    shapes, materials, warn = [], [], ""
    mat_index = {}
    cur = dict(name="", indices=[], material_ids=[])
    cur_mat = -1

    def fix(i, n):
        i = int(i)
        return i - 1 if i > 0 else n + i

    def flush():
        nonlocal cur
        if cur["indices"]:
            shapes.append(cur)

    with open(path) as f:
        for raw in f:
            t = raw.split()
            if not t or t[0].startswith("#"):
                continue
            k = t[0]
            if k == "v":
                V.extend(float(x) for x in t[1:4])
            elif k == "vn":
                VN.extend(float(x) for x in t[1:4])
            elif k == "vt":
                VT.extend([float(t[1]), float(t[2]) if len(t) > 2 else 0.0])
            elif k == "f":
                face = []
                for tok in t[1:]:
                    p = tok.split("/")
                    vi = fix(p[0], len(V) // 3)
                    ti = fix(p[1], len(VT) // 2) if len(p) > 1 and p[1] else -1
                    ni = fix(p[2], len(VN) // 3) if len(p) > 2 and p[2] else -1
                    face.append((vi, ni, ti))
                for j in range(2, len(face)):
                    cur["indices"].extend([face[0], face[j - 1], face[j]])
                    cur["material_ids"].append(cur_mat)
            elif k in ("o", "g"):
                flush()
                cur = dict(name=" ".join(t[1:]), indices=[], material_ids=[])
            elif k == "usemtl":
                cur_mat = mat_index.get(" ".join(t[1:]), -1)
            elif k == "mtllib":
                for name in t[1:]:
                    p = os.path.join(mtl_dir, name)
                    if os.path.isfile(p):
                        for m in parse_mtl(p):
                            mat_index[m["name"]] = len(materials)
                            materials.append(m)
                    else:
                        warn += "Material file [ %s ] not found.\n" % name
    flush()
    attrib = dict(vertices=np.array(V, np.float32), normals=np.array(VN, np.float32), texcoords=np.array(VT, np.float32))
    return attrib, shapes, materials, warn


def load_geometry(path, mtl_dir=None, texture_index_of=None):
    """asset_load_system.cpp:69-154 + 162-233 -> dict of pooled numpy arrays + mesh descriptor table (uint32 [n,8])."""
    attrib, shapes, materials, warn = parse_obj(path, mtl_dir)
    tex_names = []

    def default_tex(name):
        if name not in tex_names:
            tex_names.append(name)
        return tex_names.index(name)

    texture_index_of = texture_index_of or default_tex
    texture_indices = [texture_index_of(m["diffuse_texname"]) if m["diffuse_texname"] else INVALID for m in materials]
    pos, nrm, uv, idx, meshes = [], [], [], [], []
    vcount = icount = 0
    for si, sh in enumerate(shapes):
        cache, mp, mn, mt, mi = {}, [], [], [], []
        for (vi, ni, ti) in sh["indices"]:
            key = (vi, ni, ti)
            if key in cache:
                mi.append(cache[key])
                continue
            cache[key] = len(mp) // 3
            mi.append(len(mp) // 3)
            mp.extend(attrib["vertices"][3 * vi:3 * vi + 3])
            mn.extend(attrib["normals"][3 * ni:3 * ni + 3] if ni != -1 else (0.0, 0.0, 0.0))
            mt.extend(attrib["texcoords"][2 * ti:2 * ti + 2] if ti != -1 else (0.0, 0.0))
        tex = INVALID if (not sh["material_ids"] or sh["material_ids"][0] == -1) else texture_indices[sh["material_ids"][0]]
        meshes.append([len(mp) // 3, vcount, len(mi), icount, si, tex, 0, 0])
        pos.extend(mp), nrm.extend(mn), uv.extend(mt), idx.extend(mi)
        vcount += len(mp) // 3
        icount += len(mi)
    return dict(positions=np.array(pos, np.float32), normals=np.array(nrm, np.float32), texcoords=np.array(uv, np.float32),
                indices=np.array(idx, np.uint32), meshes=np.array(meshes, np.uint32).reshape(-1, 8), materials=materials,
                shapes=shapes, texture_names=tex_names, warn=warn)
