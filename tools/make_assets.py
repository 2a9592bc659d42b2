#!/usr/bin/env python3
"""Regenerate the data assets under assets/ from the reference's asset directory.

Runs only in the build container (needs /root/reference); the outputs are committed because the
reference tree does not travel to the GPU box.

* assets/bluenoise256.rgba  raw 256x256 RGBA8 texels of reference assets/textures/bluenoise256.png
                            (decoded with PIL; SURVEY.md 8c: first texel (2,57,168,54)).  This texture
                            IS the renderer's random number generator (sampling.h:13-23).
* assets/cornell_box.obj    the Cornell box geometry, re-serialised record by record (same records,
                            same order, shortest round-trip float formatting).  The `mtllib` name
                            mismatch of the original (it names cornellbox.mtl, the file is
                            cornell_box.mtl) is data and is preserved.
* assets/cornell_box.mtl    the material table (Kd / Ks / Ns / Ke records only).
"""
import os, sys
import numpy as np
from PIL import Image

REF = "/root/reference/assets"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets")


def fmt(tok):
    # shortest decimal string that round-trips to the same float32 (sign of zero kept)
    return np.format_float_positional(np.float32(float(tok)), unique=True, trim="-")


def main():
    os.makedirs(OUT, exist_ok=True)
    img = Image.open(os.path.join(REF, "textures", "bluenoise256.png")).convert("RGBA")
    a = np.asarray(img, dtype=np.uint8)
    assert a.shape == (256, 256, 4) and tuple(a[0, 0]) == (2, 57, 168, 54)
    a.tofile(os.path.join(OUT, "bluenoise256.rgba"))

    lines = ["# Cornell box scene data (records re-serialised by tools/make_assets.py)"]
    for raw in open(os.path.join(REF, "cornell_box.obj")):
        t = raw.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] in ("v", "vn", "vt"):
            lines.append(" ".join([t[0]] + [fmt(x) for x in t[1:]]))
        else:
            lines.append(" ".join(t))
    open(os.path.join(OUT, "cornell_box.obj"), "w").write("\n".join(lines) + "\n")

    keep = ("newmtl", "Kd", "Ks", "Ns", "Ke", "map_Kd")
    lines = ["# Cornell box materials (records re-serialised by tools/make_assets.py)"]
    for raw in open(os.path.join(REF, "cornell_box.mtl")):
        t = raw.split()
        if t and t[0] in keep:
            lines.append(" ".join([t[0]] + ([fmt(x) for x in t[1:]] if t[0] != "newmtl" and t[0] != "map_Kd" else t[1:])))
    open(os.path.join(OUT, "cornell_box.mtl"), "w").write("\n".join(lines) + "\n")
    print("assets written to", os.path.normpath(OUT))


if __name__ == "__main__":
    sys.exit(main())
