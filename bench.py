#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + extension + shadow rays actually traced) of the HIP wavefront path tracer on
BASELINE.json configs[1]: cornell_box.obj, 1920x1080, 64 spp, depth 8, on N MI355X GPUs of one node.

One "step" = one complete render of that image: 64 frames (frame_count 0..63) through ray generation -> BVH traversal ->
shading/BSDF sampling -> radiance accumulate; for N > 1 the frame is sharded by 8x8 screen tiles (tile t -> rank t % N) and a
step ends with ONE gather of tile radiance to rank 0 (RCCL) + the assembly of the image there.  Total work is fixed as N
grows ("strong" scaling).  Scene, BVH and blue-noise texture are resident in HBM before the timed region.

Prints one JSON line (rank 0) with the bench contract fields plus `roofline` (dominant kernel: closest-hit traversal of
extension rays, timed with HIP events on the render stream) and `cpu_baseline` (the scalar CPU oracle on the host cores).
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPP, DEPTH = 1920, 1080, 64, 8
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6.3 TB/s is the measured achievable stream rate
TREE_SPP = 32  # frames per step of the tree-path variant = one batch (tools/tree_trace.sh and tree_pmc.sh profile the same)
BYTES_CLOSEST, BYTES_ANY, BYTES_VERTEX = 48, 36, 144  # SURVEY.md 8d algorithmic queue-stream bytes per ray / shaded vertex
# vector-instruction issue peak: one wave64 instruction per 2 cycles per SIMD, 1024 SIMDs, 2.4 GHz (MI355X_MICROARCH.md)
VALU_PEAK_WAVE_INSTS_PER_S = 1024 * 2.4e9 / 2.0
def kernel_source_sha():
    """Hash of everything the profiled kernels and the trees they walk are built from -- every .hip / .h / .cpp of csrc and the
    Makefile with its flags (a builder or a layout change moves traversal bytes just as a kernel change does): committed counter
    files are only quoted while it matches."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "capsaicin_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def committed_counters(key):
    """Per-launch HBM bytes / vector instructions of one kernel from the newest profiles/rNN_traffic.json (PMC passes of this very
    command, tools/make_traffic.py; counters cannot be collected from inside the timed process).  None when the kernel sources
    have changed since the passes were taken."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
    if not files:
        return None, None
    try:
        tj = json.load(open(files[-1]))
        if tj.get("source_sha256") != kernel_source_sha():
            return None, os.path.relpath(files[-1], ROOT) + " (stale: kernel sources changed since)"
        return tj["kernels"].get(key), os.path.relpath(files[-1], ROOT)
    except Exception:  # a malformed file must not break the bench line
        return None, None


def roofline_object(kernel_name, counters_key, kernel_bytes, launches, kernel_ms, rays, extra=None, quote_counters=True):
    """The `roofline` object of the bench contract for one kernel: achieved = algorithmic bytes per launch / average launch
    duration (HIP events on the render stream).  `bound` names the limiter the counters show; the HBM fraction stays `frac`."""
    avg_ms = kernel_ms / launches
    bytes_per_launch = kernel_bytes / launches
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    pmc, src = committed_counters(counters_key) if quote_counters else (None, None)
    traffic = pmc["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 if pmc else None
    valu = None
    if pmc and pmc.get("valu_insts_per_launch"):
        valu = {"wave_insts_per_launch": pmc["valu_insts_per_launch"], "peak_wave_insts_per_s": VALU_PEAK_WAVE_INSTS_PER_S,
                "issue_frac": pmc["valu_insts_per_launch"] / (avg_ms * 1e-3) / VALU_PEAK_WAVE_INSTS_PER_S}
    o = {"bound": "valu", "bound_note": "vector-instruction issue (+ exposed latency), not HBM: see `valu` and DESIGN.md 5; "
                                        "achieved / peak / frac are the contract's algorithmic HBM bytes against the 8 TB/s peak",
         "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": traffic, "traffic_source": src, "valu": valu, "avg_launch_ms": avg_ms, "launches": int(launches),
         "algorithmic_bytes_per_launch": bytes_per_launch, "bytes_per_ray": kernel_bytes / max(1, rays),
         "rays_per_launch": rays / launches, "kernel_mrays_per_s": rays / (kernel_ms * 1e-3) / 1e6}
    if extra:
        o.update(extra)
    return o


def cpu_baseline(budget_s=12.0):
    """The oracle (kind "port": scalar C++ restatement, BVH mode) on the same scene/camera/depth, 1 frame at a time."""
    from capsaicin_amd import capi
    from oracle import cap_oracle as O
    from oracle import obj_oracle
    bn = capi.load_bluenoise()
    g = obj_oracle.load_geometry(os.path.join(ROOT, "assets", "cornell_box.obj"))
    sc = O.Scene(g["positions"], g["normals"], g["texcoords"], g["indices"], g["meshes"])
    cam = capi.cornell_camera(WIDTH, HEIGHT)
    ocam = O.make_camera(tuple(cam.position), tuple(cam.forward), tuple(cam.right), tuple(cam.up), cam.sensor_size[0],
                         cam.sensor_size[1], cam.focal_length)
    cores = os.cpu_count() or 1
    rays, frames, t0 = 0, 0, time.time()
    while True:
        r = sc.render_frame(ocam, bn, WIDTH, HEIGHT, frames, DEPTH, flags=O.FLAG_USE_BVH, threads=cores)
        rays += sum(r["rays"])
        frames += 1
        el = time.time() - t0
        if el > budget_s or frames >= 16:
            break
    # the same tracer on ONE host thread, one frame at half the resolution per axis
    w1, h1 = WIDTH // 2, HEIGHT // 2
    cam1 = capi.cornell_camera(w1, h1)
    ocam1 = O.make_camera(tuple(cam1.position), tuple(cam1.forward), tuple(cam1.right), tuple(cam1.up), cam1.sensor_size[0],
                          cam1.sensor_size[1], cam1.focal_length)
    t1 = time.time()
    r1 = sc.render_frame(ocam1, bn, w1, h1, 0, DEPTH, flags=O.FLAG_USE_BVH, threads=1)
    el1 = time.time() - t1
    return {"value": rays / el / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d frame(s) of %dx%d depth %d (of the %d spp workload), oracle BVH mode, %d threads, %.1f s" %
                      (frames, WIDTH, HEIGHT, DEPTH, SPP, cores, el),
            "single_thread": {"value": sum(r1["rays"]) / el1 / 1e6, "unit": "Mrays/s", "cores": 1,
                              "sample": "1 frame of %dx%d depth %d, %.1f s" % (w1, h1, DEPTH, el1)}}


def load_sponza_class(r, rank=0):
    """Generates and uploads the procedural 262 k-triangle textured scene (tools/make_sponza_class.py); returns its camera."""
    import tempfile
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_sponza_class as gen
    from capsaicin_amd import capi
    tmp = tempfile.mkdtemp(prefix="sponza_class_r%d_" % rank)
    gen.write(tmp, 1.0, 256)
    geo = capi.Geometry(os.path.join(tmp, "sponza_class.obj"))
    r.upload_geometry(geo)
    for i, name in enumerate(geo.texture_names):
        raw = open(os.path.join(tmp, "textures", name), "rb").read().split(b"\n", 3)
        tw, th = (int(x) for x in raw[1].split())
        rgb = np.frombuffer(raw[3], np.uint8).reshape(th, tw, 3)
        r.upload_texture(i, np.concatenate([rgb, np.full((th, tw, 1), 255, np.uint8)], -1))
    c = gen.camera()
    f = np.float64(c["forward"]) / np.linalg.norm(c["forward"])
    right = -np.cross(f, (0, 1, 0))
    right /= np.linalg.norm(right)
    camera = capi.CameraData()
    camera.position[:] = c["position"]
    camera.forward[:] = f
    camera.right[:] = right
    camera.up[:] = np.cross(f, right)
    camera.focal_length = c["focal_length"]
    camera.sensor_size[0] = 0.036
    camera.sensor_size[1] = np.float32(0.036) * (np.float32(HEIGHT) / np.float32(WIDTH))
    return camera


def shard_cost(make_renderer, spp, depth, reps=3):
    """The compute side of the 1 -> 8 GPU curve, measured on ONE device: ms per step for shard 0 of N (tiles t = 0 mod N of every
    frame, exactly what rank 0 of N renders -- every rank's share is the same to within a tile row) and the parallel efficiency it
    predicts before a byte is exchanged, T(1) / (N * T(0 of N)).  `make_renderer()` returns a Renderer with scene, tree, camera and
    resolution set."""
    r = make_renderer()
    out = {}
    for n in (1, 2, 4, 8):
        r.set_shard(0, n)
        r.render(0, spp, depth, 0)
        r.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            r.accum_reset()
            r.render(0, spp, depth, 0)
        r.sync()
        out[n] = (time.perf_counter() - t0) / reps * 1e3
    r.close()
    return {"ms_shard0_of_N": {str(n): out[n] for n in out},
            "predicted_parallel_efficiency": {str(n): out[1] / (n * out[n]) for n in out}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP, help=argparse.SUPPRESS)  # debugging only; the contract run uses 64
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-tree-variant", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true", help=argparse.SUPPRESS)  # profiling passes: no shard_cost / post_chain sections
    ap.add_argument("--batch-paths", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--traversal", type=int, default=0, help=argparse.SUPPRESS)  # 0 auto (contract run), 1 stack, 2 exhaustive
    ap.add_argument("--scene", default="cornell", help=argparse.SUPPRESS)  # "sponza": extra line on the procedural 262 k-triangle scene
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from capsaicin_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the renderer has no CPU path")
    # CAP_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices, the gather goes
    # through host memory); the contract run uses nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("CAP_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    comm_device = "cuda" if backend == "nccl" else "cpu"

    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        r = capi.Renderer(device_index, stream.cuda_stream)
        scene_name = "cornell_box.obj"
        camera = capi.cornell_camera(WIDTH, HEIGHT)
        if args.scene == "sponza":
            # not the contract workload: BASELINE configs[3] stand-in (tools/make_sponza_class.py), tree traversal kernels
            camera = load_sponza_class(r, rank)
            scene_name = "sponza_class.obj (procedural, textured)"
        else:
            r.upload_geometry(capi.Geometry(os.path.join(ROOT, "assets", "cornell_box.obj")))
        r.upload_bluenoise(capi.load_bluenoise())
        bvh = r.build_bvh()
        r.set_resolution(WIDTH, HEIGHT)
        r.set_shard(rank, world)
        r.set_camera(camera)
        if args.batch_paths:
            r.set_batch_paths(args.batch_paths)
        r.set_traversal(args.traversal)
        tile_floats = r.tile_buffer_floats()
        tile_buf = torch.zeros(tile_floats, dtype=torch.float32, device="cuda")
        gathered = torch.zeros(tile_floats * world, dtype=torch.float32, device="cuda") if rank == 0 else None
        image = torch.zeros(WIDTH * HEIGHT * 4, dtype=torch.float32, device="cuda") if rank == 0 else None

        # The exchange of the product path: cap_comm_* (capsaicin_hip.h) -- ncclGather of tile radiance + assembly on rank 0, on
        # the render stream, below Python.  The 128-byte RCCL id travels through the process group that also carries the
        # barriers.  Before anything is timed the result of one frame is compared, on rank 0, with the same gather done by
        # torch.distributed; every rank then uses the C-ABI path, or -- if RCCL could not be loaded or the comparison failed --
        # every rank uses torch.distributed, and the JSON line says which.
        exchange = "none" if world == 1 else "torch.distributed.gather"
        if world > 1 and backend == "nccl":
            # Agreement between the ranks travels over a gloo side group, on host memory: it does not queue behind a collective
            # that one rank entered and another did not, which is exactly the situation it has to detect (a rank that
            # raised inside the try skipped dist.gather while its peers waited in it).
            ctl = dist.new_group(backend="gloo")

            def agree(flag):
                t = torch.tensor([1 if flag else 0], dtype=torch.int32)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=ctl)
                return bool(t.item())

            def attempt(what, fn):
                try:
                    fn()
                    return True
                except capi.CapError as exc:
                    sys.stderr.write("[bench] rank %d: %s failed: %s\n" % (rank, what, exc))
                    return False

            try:
                ids = [capi.comm_unique_id() if rank == 0 else None]
            except capi.CapError:
                ids = [None]
            dist.broadcast_object_list(ids, src=0)
            use_cap = ids[0] is not None
            # every rank must have its communicator before any rank enters the gather
            use_cap = use_cap and agree(attempt("cap_comm_init_rank", lambda: r.comm_init_rank(ids[0], rank, world)))
            if use_cap:
                # 1. the local render (no collective): agree before anybody enters one
                def local():
                    r.accum_reset()
                    r.render(0, 1, DEPTH, 0)
                    r.sync()
                use_cap = agree(attempt("render", local))
            if use_cap:
                # 2. the C-ABI gather: only queued here; a rank that could not queue it tells the others, who then give their
                #    communicator up without waiting for the stream (cap_comm_abort) instead of hanging in cap_sync
                queued = agree(attempt("cap_comm_gather_frame", r.comm_gather_frame))
                if not queued:
                    attempt("cap_comm_abort", r.comm_abort)
                    use_cap = False
            if use_cap:
                # 3. the same tiles through torch.distributed, compared on rank 0 with what the C ABI assembled
                staged = agree(attempt("cap_resolve_tiles", lambda: (r.resolve_tiles(tile_buf.data_ptr()), r.sync())))
                same = False
                if staged:
                    torch.cuda.synchronize()
                    dist.gather(tile_buf, list(gathered.chunk(world)) if rank == 0 else None, dst=0)
                    if rank == 0:
                        def compare():
                            nonlocal same
                            r.assemble_tiles(gathered.data_ptr(), world, image.data_ptr())
                            r.sync()
                            torch.cuda.synchronize()
                            a = r.comm_readback().reshape(-1)
                            same = bool(np.array_equal(a.view(np.uint32), image.cpu().numpy().view(np.uint32)))
                        attempt("assemble / compare", compare)
                    else:
                        same = True
                use_cap = agree(staged and same)
            if use_cap:
                exchange = "cap_comm_gather_frame (ncclGather, C ABI)"
            elif ids[0] is not None:
                attempt("cap_comm_destroy", r.comm_destroy)

        def step(flags=0):
            r.accum_reset()
            r.render(0, args.spp, DEPTH, flags)
            if exchange.startswith("cap_comm"):
                r.comm_gather_frame()
                return
            r.resolve_tiles(tile_buf.data_ptr())
            if world > 1:
                # the single data-path collective: tile radiance -> rank 0 over xGMI
                if backend == "nccl":
                    dist.gather(tile_buf, list(gathered.chunk(world)) if rank == 0 else None, dst=0)
                else:
                    r.sync()
                    host = tile_buf.cpu()
                    parts = [torch.zeros_like(host) for _ in range(world)] if rank == 0 else None
                    dist.gather(host, parts, dst=0)
                    if rank == 0:
                        gathered.copy_(torch.cat(parts))
            if rank == 0:
                r.assemble_tiles((gathered if world > 1 else tile_buf).data_ptr(), world, image.data_ptr())

        def fence():
            r.sync()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()

        for _ in range(args.warmup):
            step()
        fence()
        r.stats_reset()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        st = r.stats()
        if st.guard_shade or st.guard_trace_any:
            raise SystemExit("kernel bounds guards fired: shade=%d trace_any=%d last=0x%x" % (st.guard_shade, st.guard_trace_any, st.guard_last))

        # whole-job numbers: MAX time over ranks, SUM of rays over ranks
        red = torch.tensor([dt], dtype=torch.float64, device=comm_device)
        cnt = torch.tensor([st.rays_primary, st.rays_extension, st.rays_shadow, st.shaded_vertices], dtype=torch.float64, device=comm_device)
        if world > 1:
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dt = float(red.item())
        rays_p, rays_e, rays_s, verts = (float(x) for x in cnt.tolist())
        rays = rays_p + rays_e + rays_s

        # dominant kernel, timed live with HIP events on the render stream (one extra step with per-kernel event brackets)
        r.stats_reset()
        step(capi.RENDER_STAGE_TIMERS)
        fence()
        sp = r.stats()
        roofline = None
        if rank == 0 and sp.launches_trace_closest:
            fused = sp.launches_shade == 0
            SH_ENTRY = 32  # reference-model shadow entry: (origin, path id) + (contribution, -); see DESIGN.md "Data layout"
            if fused:
                # k_trace_shade (bounce >= 1): reads one 48-B queue entry (ray 32 + throughput/path id 16) per extension ray,
                # writes one 48-B extension entry per ray it emits and one 32-B shadow entry per shadow ray its own probe does not
                # answer (CapStats::shadow_entries; DESIGN.md "algorithmic bytes")
                ext_out = sp.rays_extension - sp.rays_extension_bounce0
                sh_out = sp.shadow_entries - sp.shadow_entries_bounce0
                kernel_bytes = 48 * sp.rays_extension + 48 * ext_out + SH_ENTRY * sh_out
                kernel_name, key = "k_trace_shade<bounce>=1> (exhaustive closest hit + shading, fused)", "headline"
                # + bounce-0 kernel (3 planes + its queue writes), any-hit (16-B origin read per ray, 16-B contribution read and 12 B
                # added per unoccluded ray: counted as 16 + 12 per ray, an upper bound), resolve (48 B per path)
                all_bytes = kernel_bytes + 48 * sp.rays_extension_bounce0 + SH_ENTRY * sp.shadow_entries_bounce0 + 48 * sp.rays_primary + \
                    (16 + 12) * sp.shadow_entries + 48 * sp.rays_primary
            else:
                kernel_bytes = BYTES_CLOSEST * sp.rays_extension
                kernel_name, key = "k_trace_closest8", "tree"
                all_bytes = BYTES_CLOSEST * sp.rays_extension + 16 * sp.rays_primary + BYTES_ANY * sp.rays_shadow + BYTES_VERTEX * sp.shaded_vertices
            # the empirical stream peak of this box, measured in the same run (1 GiB float copy, read + write)
            empirical = None
            if world == 1:
                try:
                    src_t = torch.empty(1 << 28, dtype=torch.float32, device="cuda").fill_(1.0)
                    dst_t = torch.empty_like(src_t)
                    dst_t.copy_(src_t)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        dst_t.copy_(src_t)
                    e1.record()
                    torch.cuda.synchronize()
                    empirical = 5 * 2 * src_t.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
                    del src_t, dst_t
                except Exception:
                    empirical = None
            quote = args.spp == SPP and world == 1 and (args.scene == "cornell") == fused
            roofline = roofline_object(kernel_name, key, kernel_bytes, sp.launches_trace_closest, sp.ms_trace_closest, sp.rays_extension,
                                       quote_counters=quote)
            roofline.update({"empirical_stream_peak": empirical,
                             "frac_of_empirical": (roofline["achieved"] / empirical) if empirical else None,
                             "whole_step_queue_stream_gbs": all_bytes / (sp.ms_total * 1e-3) / 1e9,
                             "stage_ms": {"primary": sp.ms_primary, "trace_closest": sp.ms_trace_closest, "trace_any": sp.ms_trace_any,
                                          "shade": sp.ms_shade, "resolve": sp.ms_resolve, "total": sp.ms_total}})

        # Second line of the same workload with the EXT shading model (no reference counterpart) -- the literal "Lambert+GGX" of
        # BASELINE configs[1]: the Cornell box with its MTL colours, GGX on the two boxes and the back wall
        # (assets/scene_config.json), the emissive lamp sampled by next-event estimation.
        ext_variant = None
        if args.scene == "cornell" and world == 1:
            import shutil
            import tempfile
            tmp = tempfile.mkdtemp(prefix="cornell_mtl_")
            txt = open(os.path.join(ROOT, "assets", "cornell_box.obj")).read().replace("mtllib cornellbox.mtl", "mtllib cornell_box.mtl")
            open(os.path.join(tmp, "c.obj"), "w").write(txt)
            shutil.copy(os.path.join(ROOT, "assets", "cornell_box.mtl"), os.path.join(tmp, "cornell_box.mtl"))
            mats = capi.Geometry(os.path.join(tmp, "c.obj")).materials()
            for m, (rough, ks) in capi.scene_config()["cornell_ggx"].items():
                mats[int(m), 3], mats[int(m), 4:7] = rough, ks
            r.upload_materials(mats)
            step(capi.RENDER_EXT_MATERIALS)
            fence()
            r.stats_reset()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(capi.RENDER_EXT_MATERIALS)
            fence()
            edt = time.perf_counter() - t0
            es = r.stats()
            erays = es.rays_primary + es.rays_extension + es.rays_shadow
            r.stats_reset()
            step(capi.RENDER_EXT_MATERIALS | capi.RENDER_STAGE_TIMERS)
            fence()
            ep = r.stats()
            # EXT entries: extension 48 B + the 16 B of radiance the path has gathered (the next-event shadow ray is traced where it is
            # generated: no shadow entries, CapStats::shadow_entries == 0), read once and written once per ray that continues
            ebytes = 64 * ep.rays_extension + 64 * (ep.rays_extension - ep.rays_extension_bounce0) + 48 * (ep.shadow_entries - ep.shadow_entries_bounce0)
            ext_variant = {"workload": "cornell_box.obj + cornell_box.mtl %dx%d %dspp depth=%d, Lambert+GGX, emissive lamp + NEE (EXT model)" %
                                       (WIDTH, HEIGHT, args.spp, DEPTH),
                           "value": erays / edt / 1e6, "unit": "Mrays/s", "ms_per_step": edt / args.steps * 1e3,
                           "rays_per_step": {"primary": es.rays_primary / args.steps, "extension": es.rays_extension / args.steps,
                                             "shadow": es.rays_shadow / args.steps},
                           "roofline": roofline_object("k_trace_shade<bounce>=1, EXT> (exhaustive closest hit + GGX/NEE shading, fused)", "ext",
                                                       ebytes, ep.launches_trace_closest, ep.ms_trace_closest, ep.rays_extension,
                                                       quote_counters=args.spp == SPP),
                           "stage_ms": {"primary": ep.ms_primary, "trace_closest": ep.ms_trace_closest, "trace_any": ep.ms_trace_any,
                                        "resolve": ep.ms_resolve, "total": ep.ms_total}}

        # the general path (tree traversal, textures) on the BASELINE configs[3] stand-in, TREE_SPP spp per step (one batch of the
        # default size: configs[3]'s 128 spp are four of them): extra line, N = 1 only.
        # These are the BVH-traversal-bound frames of the north star: its roofline object is for the closest-hit kernel.
        tree_variant = None
        if world == 1 and args.scene == "cornell" and args.spp == SPP and not args.no_tree_variant:
            try:
                r2 = capi.Renderer(device_index, stream.cuda_stream)
                cam2 = load_sponza_class(r2)
                r2.upload_bluenoise(capi.load_bluenoise())
                bi2 = r2.build_bvh()
                r2.set_resolution(WIDTH, HEIGHT)
                r2.set_camera(cam2)
                r2.render(0, TREE_SPP, DEPTH, 0)
                r2.sync()
                r2.stats_reset()
                t0 = time.perf_counter()
                for _ in range(2):
                    r2.accum_reset()
                    r2.render(0, TREE_SPP, DEPTH, 0)
                r2.sync()
                tdt = time.perf_counter() - t0
                ts = r2.stats()
                trays = ts.rays_primary + ts.rays_extension + ts.rays_shadow
                r2.stats_reset()
                r2.accum_reset()
                r2.render(0, TREE_SPP, DEPTH, capi.RENDER_STAGE_TIMERS)
                r2.sync()
                tp = r2.stats()
                # (A) queue stream: 32-B ray read + 16-B hit written per ray; (B) traversal bytes per ray = nodes
                # visited x 80 B + triangles tested x 64 B from the instrumented build (tools/w8_counts.py), committed with the
                # counter passes
                pmc, _ = committed_counters("tree")
                trav = pmc.get("traversal_bytes_per_ray") if pmc else None
                troof = roofline_object("k_trace_closest8 (extension rays, compressed 8-wide tree)", "tree", BYTES_CLOSEST * tp.rays_extension,
                                        tp.launches_trace_closest, tp.ms_trace_closest, tp.rays_extension)
                troof["bound_note"] = ("texture-address path + vector-instruction issue + exposed latency (DESIGN.md 5): the nodes and "
                                       "triangles a ray touches come from L2 / Infinity Cache, not HBM; achieved / frac = queue-stream bytes "
                                       "(A), achieved_with_traversal_bytes = (A + B) requested bytes, traffic = measured HBM bytes")
                if trav:
                    ab = (BYTES_CLOSEST + trav) * tp.rays_extension / (tp.ms_trace_closest * 1e-3) / 1e9
                    troof.update({"traversal_bytes_per_ray": trav, "achieved_with_traversal_bytes": ab,
                                  "frac_with_traversal_bytes": ab / HBM_PEAK_GBS})
                # the stand-alone shade stage of this path IS bound by HBM: SURVEY.md 8d's 144 B per shaded vertex (hit 16 + path state
                # 32 read, path state 32 + extension ray 32 + shadow ray 32 written) -- the shading record of the hit triangle (96 B,
                # random) and the texels come on top and show in `traffic` when the counters are quoted
                spmc, ssrc = committed_counters("tree_shade")
                shade_ms, shade_launches = tp.ms_shade, max(1, tp.launches_shade)
                # bounce 0 is shaded by the camera-ray kernel (k_primary_shade): k_shade sees the vertices of bounces >= 1 -- at
                # least shaded_vertices - rays_primary of them (not every camera ray finds a vertex: a lower bound, on purpose)
                shade_vertices = max(0, tp.shaded_vertices - tp.rays_primary)
                sroof = {"bound": "hbm", "kernel": "k_shade, bounces >= 1 (attributes, material, direct light, BSDF sample, queue compaction)",
                         "achieved": BYTES_VERTEX * shade_vertices / (shade_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "traffic": (spmc["hbm_bytes_per_launch"] / (shade_ms / shade_launches * 1e-3) / 1e9) if spmc else None,
                         "traffic_source": ssrc, "avg_launch_ms": shade_ms / shade_launches, "launches": int(shade_launches),
                         "bytes_per_vertex": BYTES_VERTEX, "vertices_per_step": int(shade_vertices)}
                sroof["frac"] = sroof["achieved"] / HBM_PEAK_GBS
                # 56 % of k_shade's items are extension rays that escaped: each costs its 32 B of queue stream (hit 16 + throughput /
                # path id 16) and a 16-B load-add-store of the path's plane entry for the sky term (rt_indirect.hlsl:94-99) without
                # being a vertex -- compulsory bytes of this design that SURVEY.md 8d's per-vertex figure does not count
                escapes = max(0, int(tp.rays_extension) - int(shade_vertices))
                sroof["escapes_per_step"] = escapes
                sroof["achieved_with_escapes"] = (BYTES_VERTEX * shade_vertices + 64 * escapes) / (shade_ms * 1e-3) / 1e9
                sroof["frac_with_escapes"] = sroof["achieved_with_escapes"] / HBM_PEAK_GBS
                tree_variant = {"workload": "sponza_class.obj (procedural, %d triangles, textured) %dx%d %dspp depth=%d, reference shading" %
                                            (bi2.triangle_count, WIDTH, HEIGHT, TREE_SPP, DEPTH),
                                "value": trays / tdt / 1e6, "unit": "Mrays/s", "ms_per_step": tdt / 2 * 1e3,
                                "bvh": {"build": "device PLOC + 8-wide collapse (cap_bvh_build AUTO)", "depth": int(bi2.max_depth), "build_ms": float(bi2.build_ms)},
                                "roofline": troof, "shade_roofline": sroof,
                                "stage_ms": {"primary": tp.ms_primary, "trace_closest": tp.ms_trace_closest, "trace_any": tp.ms_trace_any,
                                             "shade": tp.ms_shade, "resolve": tp.ms_resolve, "total": tp.ms_total}}
                r2.close()
            except Exception as exc:  # the extra line must never cost the contract line
                tree_variant = {"error": str(exc)}

        # the compute side of the scaling curve on this one device: extra key, N = 1 only
        shard_costs = None
        if world == 1 and args.scene == "cornell" and args.spp == SPP and not args.no_tree_variant and not args.no_extras:
            try:
                def make_cornell():
                    rc = capi.Renderer(device_index, stream.cuda_stream)
                    rc.upload_geometry(capi.Geometry(os.path.join(ROOT, "assets", "cornell_box.obj")))
                    rc.upload_bluenoise(capi.load_bluenoise())
                    rc.build_bvh()
                    rc.set_resolution(WIDTH, HEIGHT)
                    rc.set_camera(capi.cornell_camera(WIDTH, HEIGHT))
                    return rc

                def make_tree():
                    rt = capi.Renderer(device_index, stream.cuda_stream)
                    camt = load_sponza_class(rt)
                    rt.upload_bluenoise(capi.load_bluenoise())
                    rt.build_bvh()
                    rt.set_resolution(WIDTH, HEIGHT)
                    rt.set_camera(camt)
                    return rt
                shard_costs = {"what": "ms per step of shard 0 of N on ONE MI355X (no exchange): the compute side of the 1 -> N curve",
                               "cornell_64spp": shard_cost(make_cornell, SPP, DEPTH),
                               "sponza_class_32spp": shard_cost(make_tree, TREE_SPP, DEPTH, reps=2)}
            except Exception as exc:  # the extra key must never cost the contract line
                shard_costs = {"error": str(exc)}

        # the stage right after the path: ms per 1080p frame of the reconstruction chain, exact (bit-identical to the
        # oracle) and with CapPostSettings::fast_weights (stated tolerance); extra key, N = 1 only
        post_chain = None
        if world == 1 and args.scene == "cornell" and args.spp == SPP and not args.no_tree_variant and not args.no_extras:
            try:
                rp = capi.Renderer(device_index, stream.cuda_stream)
                rp.upload_geometry(capi.Geometry(os.path.join(ROOT, "assets", "cornell_box.obj")))
                rp.upload_bluenoise(capi.load_bluenoise())
                rp.build_bvh()
                rp.set_resolution(WIDTH, HEIGHT)
                camp = capi.cornell_camera(WIDTH, HEIGHT)
                rp.set_camera(camp)
                rp.render(0, 1, 2, capi.RENDER_AOV)
                post_chain = {"what": "Gather -> Accumulate -> BlurDisocclusion -> Blur x4 -> Combine -> TAA, ms per %dx%d frame, static camera, "
                                      "frames 3..22" % (WIDTH, HEIGHT), "algorithmic_bytes_per_frame": WIDTH * HEIGHT * 16 * 19}
                for name, fast in (("exact", 0), ("fast_weights", 1)):
                    ps = capi.PostSettings(fast_weights=fast)
                    for f in range(3):
                        rp.post_frame(ps, f, camp)
                    rp.sync()
                    t0 = time.perf_counter()
                    for f in range(3, 23):
                        rp.post_frame(ps, f, camp)
                    rp.sync()
                    post_chain[name + "_ms"] = (time.perf_counter() - t0) / 20 * 1e3
                rp.close()
            except Exception as exc:  # the extra key must never cost the contract line
                post_chain = {"error": str(exc)}

        if rank == 0:
            # sanity of the product of the timed region: finite image, every pixel accumulated spp frames
            img = (r.comm_readback() if exchange.startswith("cap_comm") else image.cpu().numpy()).reshape(HEIGHT, WIDTH, 4)
            assert np.isfinite(img).all() and (img[..., 3] == args.spp).all(), "bench image is incomplete"
            out = {"metric": "Mrays/sec (primary+secondary), cornell_box 1080p 64spp", "value": rays / dt / 1e6, "unit": "Mrays/s",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                   "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                   "config": {"workload": scene_name + " %dx%d %dspp depth=%d (reference shading: Lambert + directional light + sky), "
                                          "tile-sharded over %d GPU(s)" % (WIDTH, HEIGHT, args.spp, DEPTH, world),
                              "triangles": int(bvh.triangle_count), "bvh_depth": int(bvh.max_depth),
                              "rays_per_step": {"primary": rays_p / args.steps, "extension": rays_e / args.steps, "shadow": rays_s / args.steps},
                              "parallelism": "tiles%d" % world, "exchange": exchange},
                   "roofline": roofline, "ext_variant": ext_variant, "tree_variant": tree_variant, "shard_cost": shard_costs, "post_chain": post_chain}
            out["cpu_baseline"] = None if (args.no_cpu_baseline or world > 1) else cpu_baseline()  # rank 0, N = 1 only
            print(json.dumps(out), flush=True)
        r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
