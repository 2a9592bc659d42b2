#!/usr/bin/env python3
"""Joins tools/micro/gather_ceiling.hip's timing lines with the counter passes of its --pmc mode (tools/micro/gather_ceiling.sh):
the n-th output line of a --pmc run is the n-th k_chase dispatch of that rocprofv3 pass.  Writes gather_ceiling.json (stdout) and a
table (gather_ceiling_report.txt) with, per fetch shape: records/s, TB/s in distinct 128-byte lines, and FETCH_SIZE against the KNOWN
bytes -- the correction factor for that access pattern (MI355X_MICROARCH.md: x 2 for wide coalesced streams, others uncalibrated).
Usage: python tools/micro/gather_ceiling_report.py gpurun_out"""
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"


def lines(path):
    return [json.loads(l) for l in open(path) if l.startswith("{")] if os.path.exists(path) else []


timing = lines(os.path.join(out_dir, "gather_ceiling_timing.jsonl"))
counters = {}  # (mode, table_mb, waves) -> {counter: value}
for jl in sorted(glob.glob(os.path.join(out_dir, "gc_pmc_*.jsonl"))):
    cfgs = lines(jl)
    fs = glob.glob(os.path.join(jl[:-6], "**", "*counter_collection.csv"), recursive=True)
    if not fs or not cfgs:
        continue
    rows = {}
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        if "k_chase" in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for cfg, did in zip(cfgs, sorted(rows)):
        counters.setdefault((cfg["mode"], cfg["table_mb"], cfg["waves_per_simd"]), {}).update(rows[did])
    if len(rows) != len(cfgs):
        sys.stderr.write("%s: %d dispatches for %d configurations\n" % (jl, len(rows), len(cfgs)))

table, txt = [], []
txt.append("%-7s %5s %2s %8s %8s %8s %8s | %10s %8s %8s %7s %7s" % ("mode", "MB", "w", "ms", "Grec/s", "TB/s128", "TB/s pay", "FETCH KiB", "x128", "x64",
                                                                   "L2 hit", "TCP/ld"))
for t in timing:
    key = (t["mode"], t["table_mb"], t["waves_per_simd"])
    c = counters.get(key, {})
    row = dict(t)
    row["counters"] = c
    if c.get("FETCH_SIZE"):
        # steps differ between the timing and the --pmc run only in --quick mode: scale by the records of the pmc launch
        pm = [x for x in lines(os.path.join(out_dir, "gc_pmc_FETCH_SIZE.jsonl")) if (x["mode"], x["table_mb"], x["waves_per_simd"]) == key]
        known128, known64 = (pm[0]["known_bytes_lines128"], pm[0]["known_bytes_sectors64"]) if pm else (t["known_bytes_lines128"], t["known_bytes_sectors64"])
        fetch_bytes = c["FETCH_SIZE"] * 1024.0
        row["fetch_size_bytes"] = fetch_bytes
        row["factor_vs_lines128"] = known128 / fetch_bytes
        row["factor_vs_sectors64"] = known64 / fetch_bytes
    hit, miss = c.get("TCC_HIT_sum"), c.get("TCC_MISS_sum")
    if hit is not None and miss:
        row["l2_hit_rate"] = hit / (hit + miss)
    table.append(row)
    txt.append("%-7s %5d %2d %8.3f %8.3f %8.3f %8.3f | %10.0f %8s %8s %7s %7s" % (
        t["mode"], t["table_mb"], t["waves_per_simd"], t.get("ms_best", 0), t.get("grecords_per_s", 0), t.get("tbs_lines128", 0), t.get("tbs_payload", 0),
        c.get("FETCH_SIZE", 0), "%.3f" % row["factor_vs_lines128"] if "factor_vs_lines128" in row else "-",
        "%.3f" % row["factor_vs_sectors64"] if "factor_vs_sectors64" in row else "-", "%.3f" % row["l2_hit_rate"] if "l2_hit_rate" in row else "-",
        "%.1f" % (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (t["records_per_launch"] / 64.0 * max(1, t["payload_bytes_per_record"] // 16)))
        if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") and t["mode"] != "stream" else "-"))
open(os.path.join(out_dir, "gather_ceiling_report.txt"), "w").write("\n".join(txt) + "\n")
print(json.dumps({"what": "tools/micro/gather_ceiling.hip: dependent random record fetches, MI355X", "rows": table}, indent=1))
