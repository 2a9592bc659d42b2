#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/prof.sh (gpurun_out/prof_pmc_*).
Usage: python tools/make_traffic.py r01   (after copying the run's bench line to profiles/r01_bench.json).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950."""
import csv, glob, json, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
KERNEL = "k_trace_shade<false, false, false, true>"


def total(counter):
    f = glob.glob(os.path.join(root, "gpurun_out", "prof_pmc_" + counter, "**", "*counter_collection.csv"), recursive=True)[0]
    s, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
            s += float(r["Counter_Value"])
            n += 1
    return s, n


fetch, n = total("FETCH_SIZE")
write, n2 = total("WRITE_SIZE")
assert n == n2 and n > 0
# launches per step of this kernel: (bounces) x batches -- read from the committed bench line
bench = json.load(open(os.path.join(root, "profiles", tag + "_bench.json")))
lps = int(bench["roofline"]["launches"])  # per step
steps = n / lps
hbm = (2.0 * fetch + write) * 1024.0
out = {"kernel": "cap::" + KERNEL, "dispatches": n, "launches_per_step": lps, "steps_profiled": steps, "FETCH_SIZE_KB_sum": fetch,
       "WRITE_SIZE_KB_sum": write, "hbm_bytes_per_step": hbm / steps, "hbm_bytes_per_launch": hbm / n,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 2 --warmup 1 "
                 "--no-cpu-baseline --no-tree-variant` (tools/prof.sh); counters are KiB summed over the kernel's dispatches; FETCH_SIZE "
                 "doubled as MI355X_MICROARCH.md prescribes for gfx950; full output in profiles/%s_rocprofv3_summary.txt" % tag}
json.dump(out, open(os.path.join(root, "profiles", tag + "_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
