#!/bin/bash
# Per-launch durations of a Cornell step with very little work (1 spp and 8 spp at 1080p): T = fixed cost + work, the measurement
# behind the 77-us-per-launch finding of round 3 (kernels.hip flush_stats).  Run through gpurun.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_tiny
cat > /tmp/tiny_step.py <<PY
import os, sys
sys.path.insert(0, "$ROOT")
from capsaicin_amd import capi
r = capi.Renderer(0)
r.upload_geometry(capi.Geometry(os.path.join("$ROOT", "assets", "cornell_box.obj"))); 
r.upload_bluenoise(capi.load_bluenoise()); r.build_bvh()
for (w, h, spp) in ((64, 64, 1), (256, 256, 4), (1920, 1080, 1), (1920, 1080, 8)):
    r.set_resolution(w, h); r.set_camera(capi.cornell_camera(w, h))
    for _ in range(2):
        r.accum_reset(); r.render(0, spp, 8, 0)
    r.sync()
PY
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_tiny -- python3 /tmp/tiny_step.py > $OUT/prof_tiny.log 2>&1 || { tail -5 $OUT/prof_tiny.log; exit 1; }
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/prof_tiny/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cap::", "")[:52]
    print("%-54s dur %7.1f  gap %8.1f us  grid %s wg %s scratch %s lds %s" % (name, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0, r.get("Grid_Size_X"), r.get("Workgroup_Size_X"), r.get("Scratch_Size", r.get("Private_Segment_Size")), r.get("LDS_Block_Size")))
    prev_end = e
PY
