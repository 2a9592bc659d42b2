#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into a small per-kernel table (kept under profiles/)."""
import csv, glob, os, sys, collections, re

def short(name):
    m = re.search(r"cap::?(\w+)|cap\d*(k_\w+?)I", name)
    name = re.sub(r"\(.*", "", name)
    return name[:60]

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
# only the passes tools/prof.sh listed for its last run (gpurun merges into gpurun_out/ and never deletes: directories of earlier
# rounds may still lie there)
manifest = os.path.join(root, "prof_manifest.txt")
if not os.path.exists(manifest):
    sys.exit("no %s: run tools/prof.sh through gpurun first" % manifest)
listed = set()
for line in open(manifest):
    k, v = line.split(None, 1)
    if k == "source_sha256":
        print("# kernel sources: sha256 %s" % v.strip())
    elif k == "pass":
        listed.add(v.strip())
# kernel trace
def newest_per_dir(pattern):
    by_dir = {}
    for f in glob.glob(pattern, recursive=True):
        if os.path.relpath(f, root).split(os.sep)[0] not in listed:
            continue
        d = f.split(os.sep)[1] if root == "gpurun_out" else os.path.dirname(os.path.dirname(f))
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


for f in newest_per_dir(os.path.join(root, "prof_*_kt", "**", "*kernel_trace.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in d.values())
    print("# kernel trace: %s" % os.path.relpath(f, root))
    print("%-62s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print("%-62s %8d %12.1f %10.1f %6.1f" % (k, len(v), sum(v), sum(v) / len(v), 100 * sum(v) / tot))
# counters
for f in newest_per_dir(os.path.join(root, "prof_*_pmc_*", "**", "*counter_collection.csv")):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print("\n# counters (summed over dispatches): %s" % os.path.relpath(f, root))
    for k, c in sorted(d.items(), key=lambda kv: -max(kv[1].values())):
        print("%-62s %s" % (k, "  ".join("%s=%.4g" % kv for kv in sorted(c.items()))))
