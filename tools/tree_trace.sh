#!/bin/bash
# Per-launch durations of the tree-path kernels on the 262 k-triangle scene (kernel trace only); run through gpurun.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_tree
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_tree -- python3 $ROOT/bench.py --scene sponza --spp 32 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_tree.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/prof_tree/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if re.search(r"k_trace|k_shade|k_resolve|k_primary", r["Kernel_Name"])]
# last batch only: from the last packet/primary launch on
start = max(i for i, r in enumerate(rows) if "primary" in r["Kernel_Name"])
for r in rows[start:]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cap::", "")[:48]
    print("%-50s %8.1f us" % (name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
