#!/usr/bin/env python3
"""Regenerates tests/golden/cornell_frames.npz: small frames of the three ray passes and of the reconstruction chain computed by
the CPU oracle (oracle/, test infrastructure) on the shipped Cornell box.  They are NOT reference-renderer output (the reference
cannot run here, DESIGN.md "Oracle and parity status"); they pin the oracle itself against drift, and give the GPU tests a fixed
target next to the live comparison.  Stored as raw uint32 bit patterns.

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cap_oracle as O  # noqa: E402
from oracle import obj_oracle  # noqa: E402

W, H = 40, 24


def camera(dx=0.0):
    return O.make_camera((-0.01 + dx, 0.995, 3.4), (0.0, 0.0, -1.0), (-1.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.036,
                         float(np.float32(0.036) * (np.float32(H) / np.float32(W))), 0.035)


def main():
    bn = np.fromfile(os.path.join(ROOT, "assets", "bluenoise256.rgba"), np.uint8).reshape(256, 256, 4)
    g = obj_oracle.load_geometry(os.path.join(ROOT, "assets", "cornell_box.obj"))
    sc = O.Scene(g["positions"], g["normals"], g["texcoords"], g["indices"], g["meshes"])
    out = {}
    # ray passes: frame 3, depth 2, plain path tracing
    ref = sc.render_frame(camera(), bn, W, H, 3, 2)
    for k in ("gbuffer_geo", "direct", "albedo", "normal_depth", "indirect", "combined"):
        out["f3_d2_" + k] = ref[k].view(np.uint32)
    out["f3_d2_rays"] = np.array(ref["rays"], np.uint64)
    # the reference's own frame loop: feedback + chain, 4 frames, camera moving from frame 2 on
    chain = O.PostChain(W, H)
    prev, pnd, hist = camera(), np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.float32)
    for f in range(4):
        cam = camera(0.02 * max(0, f - 1))
        r = sc.render_frame(cam, bn, W, H, f, 2, feedback=(prev, pnd, hist))
        o = chain.frame(O.PostSettings(), f, cam, prev, r)
        out["loop_f%d_indirect" % f] = r["indirect"].view(np.uint32)
        out["loop_f%d_output" % f] = o.view(np.uint32)
        prev, pnd, hist = cam, r["normal_depth"], o
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cornell_frames.npz"), **out)
    print("wrote tests/golden/cornell_frames.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
