"""The micro-benchmarks DESIGN.md and docs/experiments.md quote (tools/micro/*.hip) must keep compiling for gfx950: they are the
evidence behind the roofline analysis (memory-system ceiling of the traversal pattern, instruction-class costs) and are rebuilt on the GPU
box every time they are run.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("name", ["gather_ceiling", "valu_cost", "half_exec"])
def test_micro_benchmark_compiles(tmp_path, name):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path / name
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-w", os.path.join(ROOT, "tools", "micro", name + ".hip"), "-o", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.exists()


def test_gather_ceiling_numbers_are_committed():
    """bench.py's north_star object quotes the ceiling and the counter factor from this file."""
    import json
    rows = json.load(open(os.path.join(ROOT, "profiles", "r05_micro", "gather_ceiling.json")))["rows"]
    r = [x for x in rows if x["mode"] == "lane5p" and x["table_mb"] == 1331 and x["waves_per_simd"] == 6]
    assert len(r) == 1 and 6.0 < r[0]["tbs_lines128"] < 8.0 and 1.9 < r[0]["factor_vs_lines128"] < 2.1
