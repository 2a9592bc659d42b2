#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the PMC passes of tools/prof.sh (gpurun_out/prof_pmc_*): per-launch HBM bytes (FETCH_SIZE doubled
as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE) and vector instructions of the three kernels bench.py prices, plus
the hash of the kernel sources the passes were taken on (bench.py quotes the numbers only while it matches).
Usage: python tools/make_traffic.py r04 [w8_counts.json [w8_counts_big.json]]   -- the optional files are the output lines of
tools/w8_counts.py on the 262 k-triangle and on the 16.8 M-triangle hall.  Every workload has its own passes (tools/prof.sh), so
a kernel that several workloads launch (k_trace_closest8: tree and big) is priced per workload."""
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
# key -> (workload whose passes are read, kernel name)
KERNELS = {"headline": ("cornell", "k_trace_shade<false, false, false, true>"), "ext": ("ext", "k_trace_shade<false, true, false, true>"),
           "tree": ("tree", "k_trace_closest8"), "tree_shade": ("tree", "k_shade<"), "tree_any": ("tree", "k_trace_any<"),
           "tree_primary": ("tree", "k_primary_shade"),
           "big": ("big", "k_trace_closest8"), "big_shade": ("big", "k_shade<"), "big_any": ("big", "k_trace_any"),
           "config3": ("config3", "k_trace_shade<false, true, false, true>"), "config5": ("config5", "k_trace_shade<false, true, false, true>")}
# every kernel of a render step (tree path): bench.py big_variant.step_traffic = the FRAME's measured bytes, not one kernel's
STEP_KERNELS = ("k_trace_closest8", "k_trace_any", "k_shade<", "k_raygen_identity", "k_resolve", "k_primary_shade", "k_trace_primary")


def listed_passes():
    m = os.path.join(root, "gpurun_out", "prof_manifest.txt")
    if not os.path.exists(m):
        sys.exit("no gpurun_out/prof_manifest.txt: run tools/prof.sh through gpurun first")
    rows = [l.split(None, 1) for l in open(m)]
    sha = [v.strip() for k, v in rows if k == "source_sha256"]
    if sha != [bench.kernel_source_sha()]:
        sys.exit("the passes in gpurun_out/ were taken on other kernel sources (%s) than the tree holds now" % sha)
    return {v.strip() for k, v in rows if k == "pass"}


def total(pass_glob, counter, kernel):
    fs = glob.glob(os.path.join(root, "gpurun_out", pass_glob, "**", "*counter_collection.csv"), recursive=True)
    fs = [f for f in fs if os.path.relpath(f, os.path.join(root, "gpurun_out")).split(os.sep)[0] in PASSES]
    if not fs:
        return 0.0, 0
    s, n = 0.0, 0
    newest = max(fs, key=os.path.getmtime)  # gpurun merges into gpurun_out/: files of earlier runs may still lie there
    for r in csv.DictReader(open(newest)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            s += float(r["Counter_Value"])
            n += 1
            DURATION_NS[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return s, n


DURATION_NS = [0.0]  # of the dispatches the last total() summed (the counter file carries every dispatch's start and end)


def total_timed(pass_glob, counter, kernel):
    DURATION_NS[0] = 0.0
    s, n = total(pass_glob, counter, kernel)
    return s, n, DURATION_NS[0]


def static_issue_costs():
    """Average issue cycles per vector instruction of the priced kernels, from their device listings (hipcc -S here) at the measured class
    costs (tools/isa_count.py): what turns a kernel's SQ_INSTS_VALU into SIMD issue cycles.  Static mix of the whole kernel -- its loop
    dominates both the listing and the execution."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(root, "tools"))
    import isa_count
    csrc = os.path.join(root, "capsaicin_amd", "csrc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fno-vectorize", "-w",
             "-I" + os.path.join(root, "include"), "--cuda-device-only", "-S"]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for src, keys in (("trace8.hip", {"k_trace_closest8": "k_trace_closest8"}),
                          ("kernels.hip", {"k_trace_shade<false, false, false, true>": "k_trace_shadeILb0ELb0ELb0ELb1E",
                                           "k_trace_shade<false, true, false, true>": "k_trace_shadeILb0ELb1ELb0ELb1E", "k_shade<": "k_shadeILb0E",
                                           "k_trace_any<": "k_trace_anyILi24E", "k_primary_shade": "k_primary_shadeILb0E"})):
            lst = os.path.join(tmp, src + ".s")
            try:
                subprocess.run(["/opt/rocm/bin/hipcc"] + flags + [os.path.join(csrc, src), "-o", lst], check=True, capture_output=True, timeout=900)
            except Exception:
                continue
            for name, key in keys.items():
                try:
                    _, tot = isa_count.kernel_blocks(lst, key)
                    out[name] = tot["cyc"] / max(1, tot["valu"])
                except StopIteration:
                    pass
    return out


PASSES = listed_passes()
ISSUE = static_issue_costs()
out = {"source_sha256": bench.kernel_source_sha(), "kernels": {},
       "method": "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE and the SQ block, each in its own run, no trace options), one workload per "
                 "run: `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras` (headline), `bench.py --only ext`, and with "
                 "CAP_NO_TWO_LANES=1 `bench.py --only tree` / `--only big` (tools/prof.sh); counters summed over the kernel's dispatches and "
                 "divided by their number; FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for "
                 "gfx950; per-kernel table in profiles/%s_rocprofv3_summary.txt" % tag}
for key, (wl, name) in KERNELS.items():
    fetch, n = total("prof_%s_pmc_FETCH_SIZE" % wl, "FETCH_SIZE", name)
    write, n2 = total("prof_%s_pmc_WRITE_SIZE" % wl, "WRITE_SIZE", name)
    valu, n3 = total("prof_%s_pmc_SQ_WAVES*" % wl, "SQ_INSTS_VALU", name)
    if not n:
        continue
    assert n == n2, (key, n, n2)
    k = {"kernel": "cap::" + name, "workload": wl, "dispatches": n, "FETCH_SIZE_KB_sum": fetch, "WRITE_SIZE_KB_sum": write,
         "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0 / n}
    if n3:
        k["valu_insts_per_launch"] = valu / n3
    # the vector ALU's busy fraction from the counters alone.  SQ_ACTIVE_INST_VALU counts quad-cycles (4 shader
    # cycles) in which a wave has a vector instruction executing, summed over the chip's 1024 SIMDs; the clock the kernel really ran
    # at is GRBM_GUI_ACTIVE / 8 XCDs / the dispatches' own duration in that pass (MI355X_MICROARCH.md "DVFS give-back"): nothing assumed.
    act, n4, ns4 = total_timed("prof_%s_pmc_SQ_WAIT_ANY*" % wl, "SQ_ACTIVE_INST_VALU", name)
    gui, n5, ns5 = total_timed("prof_%s_pmc_TA_TA_BUSY*" % wl, "GRBM_GUI_ACTIVE", name)
    if n4 and n5 and ns4 > 0 and ns5 > 0:
        clock_ghz = gui / 8.0 / ns5
        k["valu_busy"] = {"SQ_ACTIVE_INST_VALU_quadcycles_per_launch": act / n4, "launch_ns_in_that_pass": ns4 / n4,
                          "effective_clock_ghz": clock_ghz, "simds": 1024,
                          "valu_busy_frac": act * 4.0 / (ns4 * clock_ghz * 1024.0),
                          "valu_busy_frac_at_2p4_ghz": act * 4.0 / (ns4 * 2.4 * 1024.0),
                          "note": "SQ_ACTIVE_INST_VALU is summed per WAVE in quad-cycles: an instruction of the full-rate class holds its wave for one "
                                  "quad-cycle but the SIMD for two cycles, so two waves' counts overlap and the sum can exceed the SIMD's time -- an upper "
                                  "bound of the unit's occupancy.  issue_cycles_frac prices the same launches from SQ_INSTS_VALU and the measured class costs"}
        cyc = ISSUE.get(name)
        if cyc and n3:
            vs, _, ns3 = total_timed("prof_%s_pmc_SQ_WAVES*" % wl, "SQ_INSTS_VALU", name)
            k["valu_busy"]["static_issue_cycles_per_inst"] = cyc
            k["valu_busy"]["issue_cycles_frac"] = vs * cyc / (ns3 * clock_ghz * 1024.0)
            # the scalar unit's share: one per CU, it serves the four SIMDs in turn -- tools/micro/wave_chase.hip measures ~3.4-3.9
            # cycles per scalar instruction and SIMD (profiles/r06_micro/wave_chase.txt, docs/experiments.md (85)); priced at 3.6
            ss, ns_, _ = total_timed("prof_%s_pmc_SQ_WAVES*" % wl, "SQ_INSTS_SALU", name)
            if ns_:
                k["valu_busy"]["salu_insts_per_launch"] = ss / ns_
                k["valu_busy"]["scalar_issue_cycles_frac"] = ss * 3.6 / (ns3 * clock_ghz * 1024.0)
    out["kernels"][key] = k
for wl in ("big", "tree"):
    fetch = sum(total("prof_%s_pmc_FETCH_SIZE" % wl, "FETCH_SIZE", k)[0] for k in STEP_KERNELS)
    write = sum(total("prof_%s_pmc_WRITE_SIZE" % wl, "WRITE_SIZE", k)[0] for k in STEP_KERNELS)
    n = total("prof_%s_pmc_FETCH_SIZE" % wl, "FETCH_SIZE", "k_trace_closest8")[1]
    if n:
        out["kernels"][wl + "_step"] = {"workload": wl, "kernels": list(STEP_KERNELS), "closest8_dispatches": n, "FETCH_SIZE_KB_sum": fetch,
                                        "WRITE_SIZE_KB_sum": write, "bytes_per_closest_dispatch": (2.0 * fetch + write) * 1024.0 / n}
for key, arg in (("tree", 2), ("big", 3)):
    if len(sys.argv) <= arg or key not in out["kernels"] or not os.path.exists(sys.argv[arg]):
        continue
    c = json.loads(open(sys.argv[arg]).read().strip().splitlines()[-1])
    out["kernels"][key].update({"traversal_bytes_per_ray": c["traversal_bytes_per_ray"], "node_steps_per_ray": c["node_steps_per_ray"],
                                   "triangle_tests_per_ray": c["triangle_tests_per_ray"], "lanes_per_load_sequence": c["lanes_per_load_sequence"],
                                   "traversal_bytes_source": "tools/w8_counts.py on the diagnostic build (EXTRA=-DCAP_W8_COUNT)"})
json.dump(out, open(os.path.join(root, "profiles", tag + "_traffic.json"), "w"), indent=1)
if "tree" in out["kernels"]:  # the file VERDICT round 1 asked for by name: the tree path's two priced kernels on their own
    tree = {"source_sha256": out["source_sha256"], "kernel": out["kernels"]["tree"], "method": out["method"]}
    if "tree_shade" in out["kernels"]:
        tree["shade"] = out["kernels"]["tree_shade"]
    json.dump(tree, open(os.path.join(root, "profiles", tag + "_tree_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
