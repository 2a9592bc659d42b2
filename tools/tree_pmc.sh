#!/bin/bash
# PMC pass on the tree-path bench (262 k triangles): tools/tree_pmc.sh "COUNTER1 COUNTER2 ..." (<= 4 counters of one block)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_tree_pmc
timeout -k 5 200 rocprofv3 --pmc $1 --output-format csv -d $OUT/prof_tree_pmc -- python3 $ROOT/bench.py --scene sponza --spp 32 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_tree_pmc.log 2>&1 || { tail -5 $OUT/prof_tree_pmc.log; exit 1; }
python3 - <<PY
import csv, glob, re, collections
f = glob.glob("$OUT/prof_tree_pmc/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cap::", "")[:44]
    d[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k] += 1
for k, c in sorted(d.items(), key=lambda kv: -max(kv[1].values()))[:8]:
    print("%-46s %s" % (k, "  ".join("%s=%.4g" % kv for kv in sorted(c.items()))))
PY
