"""N > 1 path on CPU: two `gloo` ranks shard an image by 8x8 screen tiles exactly as bench.py does on GPUs (tile t -> rank
t % N, fixed-size tile buffers, ONE gather to rank 0, assembly there).  The radiance comes from the CPU oracle, so this checks
the host-side sharding logic (capsaicin_amd/tiles.py, the mirror of ScreenDev in csrc/cap_device.h) and the collective
plumbing; the device kernels that fill / assemble the same layout are checked against tiles.py in tests/test_parity_gpu.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, width, height, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from capsaicin_amd import tiles
    from oracle import cap_oracle as O
    from oracle import obj_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bn = np.fromfile(os.path.join(ROOT, "assets", "bluenoise256.rgba"), np.uint8).reshape(256, 256, 4)
        g = obj_oracle.load_geometry(os.path.join(ROOT, "assets", "cornell_box.obj"))
        sc = O.Scene(g["positions"], g["normals"], g["texcoords"], g["indices"], g["meshes"])
        cam = O.make_camera((-0.01, 0.995, 3.4), (0, 0, -1), (-1, 0, 0), (0, 1, 0), 0.036, 0.036 * height / width, 0.035)
        frame = sc.render_frame(cam, bn, width, height, 3, 2)
        full = frame["combined"]
        # this rank "renders" only its shard: everything outside its tiles is discarded before the exchange
        mine = torch.from_numpy(tiles.extract(full, rank, world).copy())
        assert mine.shape[0] == tiles.padded_pixels(width, height, world)
        gathered = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, gathered, dst=0)  # the single data-path collective of a frame
        ok = True
        if rank == 0:
            img = tiles.assemble([t.numpy() for t in gathered], width, height)
            ok = bool(np.array_equal(img.view(np.uint32), full.view(np.uint32)))
        # reconstruction chain on sharded renders: the four chain inputs travel plane-major in one buffer per rank
        # (cap_resolve_aov_tiles), one gather, and the root assembles them and runs the chain (cap_post_frame_gathered)
        planes = ("indirect", "direct", "albedo", "normal_depth")
        mine4 = torch.from_numpy(np.stack([tiles.extract(frame[k], rank, world) for k in planes]).copy())
        gathered4 = [torch.zeros_like(mine4) for _ in range(world)] if rank == 0 else None
        dist.gather(mine4, gathered4, dst=0)
        if rank == 0:
            asm = {k: tiles.assemble([t.numpy()[i] for t in gathered4], width, height) for i, k in enumerate(planes)}
            ok = ok and all(np.array_equal(asm[k].view(np.uint32), np.ascontiguousarray(frame[k]).view(np.uint32)) for k in planes)
            a = O.PostChain(width, height).frame(O.PostSettings(), 3, cam, cam, dict(frame, **asm))
            b = O.PostChain(width, height).frame(O.PostSettings(), 3, cam, cam, frame)
            ok = ok and bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))
            q.put(ok)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("width,height", [(64, 48), (61, 37)])
def test_two_rank_tile_gather(width, height):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, width, height, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_tile_partition_properties():
    from capsaicin_amd import tiles
    for (w, h) in ((1920, 1080), (61, 37), (8, 8), (4096, 4096)):
        for n in (1, 2, 3, 4, 8):
            seen = np.zeros((h, w), np.int32)
            sizes = set()
            for s in range(n):
                x, y, valid = tiles.pixel_table(w, h, s, n)
                sizes.add(x.size)
                np.add.at(seen, (y[valid], x[valid]), 1)
            assert len(sizes) == 1 and (seen == 1).all()  # every pixel owned exactly once, equal buffer length on all ranks
    # shard loads differ by at most one tile
    tx, ty = tiles.tile_grid(1920, 1080)
    counts = [len(range(s, tx * ty, 8)) for s in range(8)]
    assert max(counts) - min(counts) <= 1
