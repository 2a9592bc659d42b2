#!/bin/bash
# Kernel trace of one cap_bvh_build of the hall at a scale (default 8: 16.8 M triangles): where the device build's time goes.
#   gpurun -- 'bash tools/build_trace.sh 8'     -> gpurun_out/build_trace_<scale>.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SC=${1:-8}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_build_kt
cat > /tmp/build_only.py <<PY
import sys
sys.path.insert(0, "$ROOT")
import bench
from capsaicin_amd import capi
r = capi.Renderer(0)
bench.load_sponza_class(r, scale=float("$SC"))
bi = r.build_bvh()
print("triangles %d build_ms %.1f depth %d" % (bi.triangle_count, bi.build_ms, bi.max_depth))
r.close()
PY
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_build_kt -- python3 /tmp/build_only.py > $OUT/prof_build_kt.log 2>&1 || { tail -5 $OUT/prof_build_kt.log; exit 1; }
grep triangles $OUT/prof_build_kt.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/prof_build_kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("$OUT/build_trace_$SC.txt", "w") as o:
    for r in rows[:24]:
        line = "%-60s calls %5s  total %9.3f ms  avg %8.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot)
        print(line); o.write(line + "\n")
PY
