#!/bin/bash
# Counter pass over the reconstruction chain at 1080p (tools/time_post.py): tools/post_pmc.sh "COUNTER1 COUNTER2 ..." [fast]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_post_pmc
timeout -k 5 200 rocprofv3 --pmc $1 --output-format csv -d $OUT/prof_post_pmc -- python3 $ROOT/tools/time_post.py 1920 1080 20 $2 > $OUT/prof_post_pmc.log 2>&1 || { tail -5 $OUT/prof_post_pmc.log; exit 1; }
python3 - <<PY
import csv, glob, re, collections
f = glob.glob("$OUT/prof_post_pmc/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void cap::", "").replace("cap::", "")[:44]
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, c in sorted(d.items(), key=lambda kv: -max(kv[1].values()))[:9]:
    print("%-46s %s" % (k, "  ".join("%s=%.4g" % (a, b / max(1, n[(k, a)])) for a, b in sorted(c.items()))))
PY
