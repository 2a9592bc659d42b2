#!/usr/bin/env python3
"""Static instruction mix of one kernel of a device assembly listing (hipcc --cuda-device-only -S), per basic block.
    python tools/isa_count.py kernels.s 'k_trace_shadeILb0ELb1ELb0ELb1E' [min instructions per printed block]
Prints every label with its VALU / SALU / VMEM / LDS / SMEM counts and the kernel's totals: for the persistent kernels, whose
body is one loop, the sum over the loop's blocks is the instruction count per chunk (branches aside).  `cyc` prices the vector
instructions at their measured issue cost per SIMD with several waves resident (profiles/r06_micro/valu_cost.txt, docs/experiments.md
(74)): 2 cycles for the full-rate class (v_fma / v_mul / v_add / v_sub f32, v_mov, v_add_u32, v_bitop3 and the two-operand bit
operations), 8 for transcendentals, 4 for everything else (byte / integer conversions, min / max incl. the three-operand forms, selects,
compares, shifts, bit-field, count and permute operations, v_mul_lo_u32, v_addc).  Importable: kernel_blocks(path, key)."""
import re
import sys

TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")
FULL = ("v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_mov_b64",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_bitop3_b32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32")
ZERO = dict(valu=0, salu=0, vmem=0, lds=0, smem=0, trans=0, cyc=0)


def cost(op):
    if op.startswith(TRANS):
        return 8
    return 2 if op.startswith(FULL) else 4


def kernel_blocks(path, key):
    """(blocks, totals): blocks = [(label, counts)], counts / totals = instruction counts by unit + `cyc` (vector issue cycles)."""
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(key), l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = [], ["entry", dict(ZERO)]
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur)
            cur = [t.split(":")[0], dict(ZERO)]
            continue
        op = t.split()[0] if t and not t.startswith((";", ".")) else ""
        if op.startswith("v_"):
            cur[1]["valu"] += 1
            cur[1]["cyc"] += cost(op)
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
    tot = dict(ZERO)
    for _, c in blocks:
        for k in tot:
            tot[k] += c[k]
    return blocks, tot


if __name__ == "__main__":
    blocks, tot = kernel_blocks(sys.argv[1], sys.argv[2])
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    for name, c in blocks:
        if sum(c.values()) - c["cyc"] >= floor:
            print("%-12s %s" % (name, "  ".join("%s=%d" % kv for kv in c.items())))
    print("%-12s %s" % ("TOTAL", "  ".join("%s=%d" % kv for kv in tot.items())))
    print("static average: %.2f issue cycles per vector instruction" % (tot["cyc"] / max(1, tot["valu"])))
