"""The multi-GPU frame exchange below Python (include/capsaicin_hip.h cap_comm_*): one gather of tile radiance to
rank 0 + the assembly there.  On a one-GPU box this covers: RCCL itself with a one-rank communicator (library load,
ncclCommInitRank, ncclGather on the context's stream), the single-process form with several shards on one device (device
copies instead of links, same staging / assembly code), the C++ host layer driving it (capsaicin_viewer --gpus N), the ray
counters of the shards against the unsharded render, and bench.py started as two ranks.  With two or more GPUs the RCCL
communicator over distinct devices runs as well."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from capsaicin_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def make(cornell_path, bluenoise, w, h, shard=(0, 1), device=0):
    r = capi.Renderer(device)
    r.upload_geometry(capi.Geometry(cornell_path))
    r.upload_bluenoise(bluenoise)
    r.build_bvh()
    r.set_resolution(w, h)
    r.set_shard(*shard)
    r.set_camera(capi.cornell_camera(w, h))
    return r


def test_rccl_one_rank_gather(native_lib, bluenoise, cornell_path):
    w, h = 200, 120
    r = make(cornell_path, bluenoise, w, h)
    r.comm_init_rank(capi.comm_unique_id(), 0, 1)
    assert r.comm_info() == (0, 1, True)
    r.render(0, 3, 3)
    r.comm_gather_frame()
    got = r.comm_readback()
    assert np.array_equal(bits(got), bits(r.readback(capi.BUF_ACCUM_MEAN)))
    with pytest.raises(capi.CapError, match="already has a communicator"):
        r.comm_init_rank(capi.comm_unique_id(), 0, 1)
    r.comm_destroy()
    with pytest.raises(capi.CapError):
        r.comm_gather_frame()
    r.close()


@pytest.mark.parametrize("n", [2, 3, 8])
def test_shards_of_one_process(native_lib, bluenoise, cornell_path, n):
    w, h, spp, D = 203, 117, 4, 4  # partial tiles on both edges
    whole = make(cornell_path, bluenoise, w, h)
    whole.render(0, spp, D)
    want = whole.readback(capi.BUF_ACCUM_MEAN)
    ws = whole.stats()
    shards = [make(cornell_path, bluenoise, w, h, (i, n)) for i in range(n)]
    capi.comm_init_all(shards)
    assert all(s.comm_info() == (i, n, False) for i, s in enumerate(shards))  # one device: copies, no communicator
    for s in shards:
        s.render(0, spp, D)
    capi.comm_gather_frame_all(shards)
    got = shards[0].comm_readback()
    assert np.array_equal(bits(got), bits(want))
    # every ray is traced by exactly one shard
    st = [s.stats() for s in shards]
    for field in ("rays_primary", "rays_extension", "rays_shadow", "shaded_vertices"):
        assert sum(getattr(x, field) for x in st) == getattr(ws, field), field
    with pytest.raises(capi.CapError, match="rank 0"):
        shards[1].comm_readback()
    for s in shards + [whole]:
        s.close()


def test_mismatched_shard_count_is_an_error_and_recoverable(native_lib, bluenoise, cornell_path):
    """A context whose shard does not match its rank (cap_set_shard after cap_comm_init_*) must fail with a status, leave no RCCL
    group open and no half-staged frame behind: the same contexts gather correctly once the shards are put right.  Both forms:
    the one-process contexts of cap_comm_init_all and a one-rank RCCL communicator (cap_comm_init_rank)."""
    w, h, spp, D, n = 203, 117, 2, 3, 3
    whole = make(cornell_path, bluenoise, w, h)
    whole.render(0, spp, D)
    want = whole.readback(capi.BUF_ACCUM_MEAN)
    shards = [make(cornell_path, bluenoise, w, h, (i, n)) for i in range(n)]
    capi.comm_init_all(shards)
    shards[2].set_shard(1, 2)  # deliberately wrong: renders shard 1 of 2 while being rank 2 of 3
    for s in shards:
        s.render(0, spp, D)
    with pytest.raises(capi.CapError, match="renders shard 1 of 2 but is rank 2 of 3"):
        capi.comm_gather_frame_all(shards)
    with pytest.raises(capi.CapError, match="same order"):
        capi.comm_gather_frame_all(shards[:2])  # wrong count
    shards[2].set_shard(2, n)
    for s in shards:
        s.accum_reset()
        s.render(0, spp, D)
    capi.comm_gather_frame_all(shards)
    capi.comm_gather_frame_all(shards)  # twice without a cap_sync in between: the shards' send buffers are fenced by events
    assert np.array_equal(bits(shards[0].comm_readback()), bits(want))
    for s in shards:
        s.close()
    # RCCL form: rank 0 of 1, then the shard is changed under the communicator
    r = make(cornell_path, bluenoise, w, h)
    r.comm_init_rank(capi.comm_unique_id(), 0, 1)
    r.set_shard(0, 2)
    r.render(0, spp, D)
    with pytest.raises(capi.CapError, match="renders shard 0 of 2 but is rank 0 of 1"):
        r.comm_gather_frame()
    r.set_shard(0, 1)
    r.accum_reset()
    r.render(0, spp, D)
    r.comm_gather_frame()  # the async-error poll of the second frame and the collective itself: clean
    assert np.array_equal(bits(r.comm_readback()), bits(want))
    r.close()
    whole.close()


@pytest.mark.skipif(capi.device_count() < 2, reason="needs two GPUs")
def test_rccl_across_devices(native_lib, bluenoise, cornell_path):
    n = min(capi.device_count(), 4)
    w, h, spp, D = 256, 160, 3, 3
    whole = make(cornell_path, bluenoise, w, h)
    whole.render(0, spp, D)
    want = whole.readback(capi.BUF_ACCUM_MEAN)
    shards = [make(cornell_path, bluenoise, w, h, (i, n), device=i) for i in range(n)]
    capi.comm_init_all(shards)
    assert all(s.comm_info()[2] for s in shards)
    for s in shards:
        s.render(0, spp, D)
    capi.comm_gather_frame_all(shards)
    assert np.array_equal(bits(shards[0].comm_readback()), bits(want))
    for s in shards + [whole]:
        s.close()


def test_viewer_gpus_flag(native_lib, tmp_path):
    viewer = os.path.join(ROOT, "capsaicin_amd", "capsaicin_viewer")
    env = dict(os.environ, CAPSAICIN_ASSETS=os.path.join(ROOT, "assets"))
    outs, logs = [], []
    for n in (1, 3):
        out = str(tmp_path / ("g%d.ppm" % n))
        p = subprocess.run([viewer, "--scene", os.path.join(ROOT, "assets", "cornell_box.obj"), "--out", out, "--width", "168", "--height", "96",
                            "--frames", "4", "--bounces", "2", "--gpus", str(n)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(open(out, "rb").read())
        logs.append([l for l in p.stderr.splitlines() if "rays primary/extension/shadow" in l][0].split("rays primary")[1])
    assert outs[0] == outs[1], "the frame assembled from three shards differs from the unsharded one"
    assert logs[0] == logs[1], "ray counters: %s vs %s" % (logs[0], logs[1])


def test_bench_two_ranks(native_lib):
    """bench.py exactly as the driver starts it for N = 2, both ranks sharing this box's GPU (CAP_BENCH_BACKEND=gloo rehearsal:
    same sharding, render and assembly code, the gather through host memory instead of RCCL)."""
    env = dict(os.environ, CAP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--spp", "4"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["config"]["parallelism"] == "tiles2"
    # the rays of the two shards add up to the unsharded count of the same 4 frames
    r = make(os.path.join(ROOT, "assets", "cornell_box.obj"), capi.load_bluenoise(), 1920, 1080)
    r.render(0, 4, 8)
    s = r.stats()
    rp = d["config"]["rays_per_step"]
    assert (rp["primary"], rp["extension"], rp["shadow"]) == (s.rays_primary, s.rays_extension, s.rays_shadow)
    r.close()
    # what a SCALE run needs from the line: which exchange the step ended with and where each rank's time went
    assert d["exchange"] == d["config"]["exchange"] == "torch.distributed.gather"  # the gloo rehearsal's path; nccl: cap_comm_gather_frame
    assert len(d["stage_ms_per_rank"]) == 2
    for st in d["stage_ms_per_rank"]:
        assert set(st) == {"primary", "trace_closest", "trace_any", "shade", "resolve", "total"} and st["total"] > 0 and st["trace_closest"] > 0
    assert d["roofline"]["stage_ms"]["total"] == d["stage_ms_per_rank"][0]["total"]


def test_bench_two_ranks_sponza_workload(native_lib):
    """--workload sponza: BASELINE configs[3] as the sharded workload (the second scaling curve), same rehearsal."""
    env = dict(os.environ, CAP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           "29519", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "sponza", "--spp", "2"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["triangles"] > 250_000 and "sponza_class" in d["config"]["workload"] and "configs[3]" in d["metric"]
    assert d["exchange"] == "torch.distributed.gather" and len(d["stage_ms_per_rank"]) == 2
    assert all(st["shade"] > 0 and st["trace_any"] > 0 for st in d["stage_ms_per_rank"])  # the tree path's stand-alone stages ran
    assert d["config"]["rays_per_step"]["primary"] == 2 * 1920 * 1080
    # every rank built its own copy of the tree (device SAH splits + clustering + 8-wide collapse): the same bytes on both
    sha = d["config"]["wide_tree_sha1_per_rank"]
    assert len(sha) == 2 and sha[0] and sha[0] == sha[1]
