"""The micro-benchmarks DESIGN.md and docs/experiments.md quote (tools/micro/*.hip) are the evidence behind the roofline analysis
(memory-system ceiling of the traversal pattern, instruction-class costs); they are rebuilt on the GPU box every time they are run.
Default suite: the instruction-cost benchmark's device listing (seconds; hipcc cross-compiles without a GPU) -- its numbers are only
worth quoting while nothing sits between the timed instructions.  CAP_TEST_MICRO=1 adds full builds of all of them (slow: ADVICE r5)."""
import json
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def test_valu_cost_listing_has_nothing_between_the_timed_instructions(tmp_path):
    """Round 5's table was wrong by a factor of two for a wave alone: hipcc had put an `s_nop 0` between every two one-instruction asm
    statements (they clobbered vcc / SGPRs).  The timed blocks are now single asm statements; the listing must show no s_nop inside a
    timed loop outside them, and a loop body that is exactly one asm block plus the loop's three scalar instructions."""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path / "valu_cost.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-w", "--cuda-device-only", "-S", os.path.join(ROOT, "tools", "micro", "valu_cost.hip"), "-o", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    in_asm = in_loop = False
    loops = 0
    for line in out.read_text().splitlines():
        t = line.strip()
        if "#ASMSTART" in t:
            in_asm = True
        elif "#ASMEND" in t:
            in_asm = False
        elif re.match(r"^\.LBB\d+_\d+:.*Loop Header", t):
            in_loop, body = True, []
            loops += 1
        elif in_loop and not in_asm and t and not t.startswith(";"):
            body.append(t.split()[0])
            if t.startswith("s_cbranch"):
                in_loop = False
                assert sorted(body) == ["s_add_i32", "s_cbranch_scc0", "s_cmp_eq_u32"] or sorted(body) == ["s_add_i32", "s_cbranch_scc1", "s_cmp_lg_u32"], body
    assert loops >= 30


@pytest.mark.skipif(os.environ.get("CAP_TEST_MICRO") != "1", reason="full builds of the micro-benchmarks: CAP_TEST_MICRO=1")
@pytest.mark.parametrize("name", ["gather_ceiling", "valu_cost"])
def test_micro_benchmark_compiles(tmp_path, name):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path / name
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-w", os.path.join(ROOT, "tools", "micro", name + ".hip"), "-o", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.exists()


def test_gather_ceiling_numbers_are_committed():
    """bench.py's north_star object quotes the ceiling and the counter factor from this file: the row must be there and carry numbers
    (what the numbers are is a measurement, not a test: ADVICE r5)."""
    rows = json.load(open(os.path.join(ROOT, "profiles", "r05_micro", "gather_ceiling.json")))["rows"]
    r = [x for x in rows if x["mode"] == "lane5p" and x["table_mb"] == 1331 and x["waves_per_simd"] == 6]
    assert len(r) == 1
    assert isinstance(r[0]["tbs_lines128"], float) and r[0]["tbs_lines128"] > 0 and isinstance(r[0]["factor_vs_lines128"], float)
