"""Step counts of the wide closest-hit kernel on the procedural hall (diagnostic build: make -C capsaicin_amd/csrc -B
EXTRA=-DCAP_W8_COUNT).  python tools/w8_counts.py [scale [spp]]  -> per-ray node steps / triangle tests, lane utilisation of the two
phases, and the traversal bytes (B) of SURVEY.md 8d per ray.  scale 1 (default): 262 k triangles, 16 spp; scale 8: 16.8 M, 8 spp."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from capsaicin_amd import capi  # noqa: E402


def main():
    lib = capi.lib()
    if not hasattr(lib, "cap_debug_w8_counts"):
        raise SystemExit("needs the diagnostic build: make -C capsaicin_amd/csrc -B EXTRA=-DCAP_W8_COUNT")
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else (16 if scale <= 1.0 else bench.BIG_SPP)
    r = capi.Renderer(0)
    cam = bench.load_sponza_class(r, scale=scale)
    r.upload_bluenoise(capi.load_bluenoise())
    r.build_bvh()
    r.set_resolution(bench.WIDTH, bench.HEIGHT)
    r.set_camera(cam)
    out = (ctypes.c_ulonglong * 16)()
    r.render(0, spp, bench.DEPTH, 0)
    r.sync()
    lib.cap_debug_w8_counts(out, 1)
    r.stats_reset()
    r.accum_reset()
    r.render(0, spp, bench.DEPTH, 0)
    r.sync()
    s = r.stats()
    lib.cap_debug_w8_counts(out, 1)
    nodes, tris, seqs, top, rays, iters, pushes, spills, cull_child, cull_group, empty = (int(x) for x in list(out)[:11])
    res = {"scale": scale, "spp": spp, "triangles": int(r.bvh_info().triangle_count), "rays": rays, "rays_extension": int(s.rays_extension), "node_steps_per_ray": nodes / rays, "triangle_tests_per_ray": tris / rays,
           "lanes_per_load_sequence": (nodes + tris) / max(1, seqs), "load_sequences": seqs, "loop_iterations": iters,
           "node_steps_on_top_levels": top / max(1, nodes), "pushes_per_ray": pushes / rays,
           "spilled_push_fraction": spills / max(1, pushes),
           "steps_a_per_child_cull_would_skip": cull_child / max(1, nodes), "steps_a_group_bound_would_skip": cull_group / max(1, nodes),
           "steps_that_hit_no_child": empty / max(1, nodes),
           # nodes visited x node size + triangles tested x record size
           "traversal_bytes_per_ray": 80 * nodes / rays + 64 * tris / rays}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
