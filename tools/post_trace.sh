#!/bin/bash
# Per-kernel durations of the reconstruction chain at 1080p (kernel trace of tools/time_post.py); run through gpurun.  POST_MODE=fast: CapPostSettings::fast_weights
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_post
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_post -- python3 $ROOT/tools/time_post.py 1920 1080 20 $POST_MODE > $OUT/prof_post.log 2>&1 || { tail -5 $OUT/prof_post.log; exit 1; }
tail -1 $OUT/prof_post.log
python3 - <<PY
import csv, glob, re, collections
f = glob.glob("$OUT/prof_post/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void cap::", "")[:50]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-52s %5d calls  %9.1f us total  %8.1f us avg" % (k, len(v), sum(v), sum(v) / len(v)))
PY
