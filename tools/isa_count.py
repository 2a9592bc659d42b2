#!/usr/bin/env python3
"""Static instruction mix of one kernel of a device assembly listing (hipcc --cuda-device-only -S), per basic block.
    python tools/isa_count.py kernels.s 'k_trace_shadeILb0ELb1ELb0ELb1E'
Prints every label with its VALU / SALU / VMEM / LDS / SMEM counts and the kernel's totals: for the persistent kernels, whose
body is one loop, the sum over the loop's blocks is the instruction count per chunk (branches aside)."""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(key), l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, cur = [], ["entry", dict(valu=0, salu=0, vmem=0, lds=0, smem=0, trans=0)]
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append(cur)
        cur = [t.split(":")[0], dict(valu=0, salu=0, vmem=0, lds=0, smem=0, trans=0)]
        continue
    op = t.split()[0] if t and not t.startswith((";", ".")) else ""
    if op.startswith("v_"):
        cur[1]["valu"] += 1
        if op.startswith(TRANS):
            cur[1]["trans"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cur[1]["vmem"] += 1
    elif op.startswith("ds_"):
        cur[1]["lds"] += 1
    elif op.startswith("s_load") or op.startswith("s_buffer_load"):
        cur[1]["smem"] += 1
    elif op.startswith("s_"):
        cur[1]["salu"] += 1
blocks.append(cur)
tot = dict(valu=0, salu=0, vmem=0, lds=0, smem=0, trans=0)
for name, c in blocks:
    if sum(c.values()) >= (int(sys.argv[3]) if len(sys.argv) > 3 else 20):
        print("%-12s %s" % (name, "  ".join("%s=%d" % kv for kv in c.items())))
    for k in tot:
        tot[k] += c[k]
print("%-12s %s" % ("TOTAL", "  ".join("%s=%d" % kv for kv in tot.items())))
