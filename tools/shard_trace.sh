#!/bin/bash
# Per-launch durations and gaps of ONE step of shard 0 of N (kernel trace): tools/shard_trace.sh <N> [sponza]; run through gpurun.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
N=${1:-8}
SCENE=${2:-cornell}
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_shard
cat > /tmp/shard_step.py <<PY
import os, sys
sys.path.insert(0, "$ROOT")
import bench
from capsaicin_amd import capi
r = capi.Renderer(0)
if "$SCENE" == "sponza":
    cam = bench.load_sponza_class(r); spp = bench.TREE_SPP
else:
    r.upload_geometry(capi.Geometry(os.path.join("$ROOT", "assets", "cornell_box.obj"))); cam = capi.cornell_camera(1920, 1080); spp = 64
r.upload_bluenoise(capi.load_bluenoise()); r.build_bvh(); r.set_resolution(1920, 1080); r.set_camera(cam); r.set_shard(0, $N)
for _ in range(3):
    r.accum_reset(); r.render(0, spp, 8, 0)
r.sync()
PY
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_shard -- python3 /tmp/shard_step.py > $OUT/prof_shard.log 2>&1 || { tail -5 $OUT/prof_shard.log; exit 1; }
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/prof_shard/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if re.search(r"cap::", r["Kernel_Name"])]
# the last step: from the last bounce-0 launch on
first = [i for i, r in enumerate(rows) if re.search(r"k_trace_shade<true|k_primary_shade", r["Kernel_Name"])][-1]
rows = rows[first:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0; busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cap::", "")[:52]
    print("%-54s start %8.1f  dur %7.1f  gap %6.1f us" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    busy += e - s; prev_end = e
print("span %.1f us, kernels %.1f us, %d launches" % ((prev_end - t0) / 1e3, busy / 1e3, len(rows)))
PY
