import sys, os, tempfile, subprocess, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from capsaicin_amd import capi
d = tempfile.mkdtemp()
subprocess.check_call([sys.executable, "tools/make_sponza_class.py", d], stdout=subprocess.DEVNULL)
geo = capi.Geometry(os.path.join(d, "sponza_class.obj"))
for mode in (1, 2):
    r = capi.Renderer(0)
    r.upload_geometry(geo)
    r.set_bvh_build(mode)
    bi = r.build_bvh()
    bi = r.build_bvh()
    print("mode", mode, "tris", bi.triangle_count, "depth", bi.max_depth, "stack", bi.stack_entries, "build ms %.2f" % bi.build_ms)
    r.close()
