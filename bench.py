#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s, path-traced ~1M-triangle scene at 1920x1080 (BASELINE.json configs[2]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" is one Renderer::Tick-equivalent pass of the hot path over one batch: every rank
path-traces --spp-per-step samples for every pixel (ray generation, extend, shade, connect, ... until every path
has terminated) and adds them into its float4 accumulator in HBM.  Samples are split by index over the ranks
(rank r renders sample indices r, r+N, r+2N, ...), so per-GPU work is fixed as N grows (weak scaling:
N GPUs deliver N x K x spp-per-step samples per pixel).  At the end of the timed region the accumulators are
summed onto rank 0 with ONE RCCL reduce (torch.distributed, backend nccl) and rank 0 converts to RGB8 on device.
Scene, BVH, camera, seed table and all queues are resident in HBM before the timed region starts.

value = (closest-hit + any-hit rays traversed by all ranks, from device counters) / (max over ranks of the time of
the K steps + reduce + finalize), in Mrays/s.  The reference's own "Mrays/s" (primary samples/s, renderer.cpp:639)
is reported beside it as mpaths_per_s.

roofline: the dominant kernel is k_trace<closest> (closest-hit BVH walk of the rays that entered the root box).
achieved = algorithmic bytes of all its launches in the timed region / their summed duration measured with hipEvents
recorded on the stream the kernels run on (inside librtmi, rtmi_set_profiling), i.e. bytes per launch / mean launch
duration.  Algorithmic bytes per walked ray = 64 B x node-pair records fetched + 48 B x leaf primitive records tested
+ 44 B of ray state (32 B read, 4 B walk index, 8 B hit write); the per-walker record counts
come from a counting pass of the same kernels over the first step's samples.  Since a bounce's k_trace<any> runs beside
the next bounce's k_trace<closest> (side stream), each launch's duration includes the other's share of the chip;
roofline.both_walks gives both kernels' algorithmic bytes over the windows they share (k_shade to k_shade).

cpu_baseline: the CPU restatement (oracle/, kind "port"; validated bit-exact against the compiled reference) timed
on this box's host cores over a bounded row subset of one sample of the same workload.  N=1, rank 0 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=64, help="samples per pixel per step per GPU (configs[2]: 64 spp progressive)")
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--builder", choices=["binned", "reference"], default="binned")
    ap.add_argument("--zoom", type=float, default=1.0, help="NOT the headline: shrink the screen rectangle about its centre (a view filled by the middle knot)")
    ap.add_argument("--tube", type=float, default=0.075, help="NOT the headline: tube radius of the knots (0.075 in every config)")
    ap.add_argument("--hw-queues", type=int, default=16,
                    help="GPU_MAX_HW_QUEUES for this process (HIP's default is 4; 0 = leave the environment alone).  With 12 or more the library "
                         "renders on two lanes (two batches in flight side by side, option lanes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--traffic-bytes-per-launch", type=float, default=None,
                    help="HBM bytes per k_trace<closest> launch; default: the committed result of the separate rocprofv3 --pmc "
                         "passes (profiles/rNN_traffic.json, tools/profile_gpu.sh + tools/traffic.py), null if absent")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the headline): every rank renders --spp-per-step samples per step.  strong: a step is ONE frame of "
                         "--total-spp samples split by index over the ranks (BASELINE.json configs[3]: 256 spp at 3840x2160 over 2/4/8 GPUs)")
    ap.add_argument("--total-spp", type=int, default=256, help="--scaling strong: samples per pixel per step, all ranks together")
    ap.add_argument("--share-of", type=int, default=1,
                    help="single process only: render the share of rank 0 of this many ranks (sample indices 0, N, 2N, ...), i.e. what ONE GPU of an "
                         "N-GPU run does between the start and the reduce; the JSON line is then a per-rank rehearsal, not a whole-node figure")
    ap.add_argument("--reduce", choices=["torch", "rtmi"], default="torch",
                    help="who sums the accumulators: torch.distributed (default) or rtmi_reduce_accumulator on an RCCL communicator of its own")
    return ap.parse_args()


def committed_pmc():
    """PMC counters cannot be collected inside this run (rocprofv3 --pmc is its own pass per counter group): the figures of the
    committed passes of the same command (tools/profile_gpu.sh -> tools/traffic.py), newest round first.  Returns (dict, file)."""
    for name in ("r02_traffic.json", "r02_lanes1_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                k = json.load(f)["kernels"]["k_trace<closest>"]
            return k, "profiles/" + name
        except (OSError, KeyError, ValueError):
            continue
    return {}, None


def main():
    a = parse()
    # hardware queues: the library's streams (per lane: main, side, four tails) must not share them; the HIP runtime reads this
    # when it starts, i.e. at the first import of torch below
    if a.hw_queues > 0 and int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < a.hw_queues:
        os.environ["GPU_MAX_HW_QUEUES"] = str(min(a.hw_queues, 32))
    pmc, pmc_file = committed_pmc()
    traffic_source = "--traffic-bytes-per-launch" if a.traffic_bytes_per_launch is not None else None
    if a.traffic_bytes_per_launch is None and "traffic_bytes_per_launch" in pmc:
        a.traffic_bytes_per_launch = float(pmc["traffic_bytes_per_launch"])
        traffic_source = "%s: separate rocprofv3 --pmc passes of an earlier run of this command (%s), not this run" % (pmc_file, pmc.get("command", "bench.py --steps 3"))
    import torch
    import torch.distributed as dist
    from raytracer_amd import api
    from raytracer_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace loop has no CPU path (see oracle/ for the checker)")
    torch.cuda.set_device(local_rank)
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run (also with one rank: same code path as N ranks)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    W, H = a.width, a.height
    t0 = time.time()
    scene = api.mesh_scene(a.tris, api.BINNED_SAH if a.builder == "binned" else api.REFERENCE_SAH, tube=a.tube)
    info = scene.info()
    build_s = time.time() - t0
    dev = api.Device(W, H, local_rank)
    dev.upload(scene.desc)
    dev.set_camera(api.zoom_camera(api.default_camera(W, H, scene.scale), a.zoom))
    # how much of the view is mesh: camera rays whose nearest hit is a triangle (the analytic primitives come first in the numbering)
    n_analytic = info["quads"] + info["spheres"] + info["cubes"] + info["planes"]
    primary_mesh_fraction = float((dev.trace_primary()[0] >= n_analytic).mean())
    seeds = api.seed_table(W * H)
    dev.set_seeds(seeds)
    # Everything below runs on ONE explicit torch stream: the library enqueues on it (rtmi_set_stream takes the stream as it
    # is), rtmi_flush makes it wait for the library's internal streams, and the collective -- issued while it is torch's current
    # stream -- is ordered after it by torch's own event.  (On torch's default stream the handle would get the null stream;
    # that works too, but a named stream keeps the ordering explicit and independent of null-stream semantics.)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    dev.set_stream(stream.cuda_stream)
    dev.set_accumulator(acc.data_ptr())
    strong = a.scaling == "strong"
    share = max(1, a.share_of)
    if share > 1 and world > 1:
        raise SystemExit("--share-of is a single-process rehearsal")
    v_world, v_rank = (share, 0) if share > 1 else (world, rank)  # the split this process renders its part of
    # samples per step over all (virtual) ranks: weak = spp-per-step each, strong = one frame of total-spp
    step_spp = a.total_spp if strong else a.spp_per_step * v_world
    spp = a.spp_per_step
    comm = None
    if a.reduce == "rtmi" and use_dist:
        comm = rccl_communicator(dist, rank, world, local_rank)

    def step(i):
        # step i covers the global sample indices [i*step_spp, (i+1)*step_spp); this rank renders every v_world-th one
        D.render_split(lambda b, n, st: dev.render(api.PATHTRACING, b, n, st), i * step_spp, step_spp, v_rank, v_world)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # counting pass (not timed): node-pair / primitive records per ray on this workload
    dev.set_profiling(False, True)
    dev.reset_stats()
    step(0)
    dev.sync()
    cs = dev.stats()
    pairs_per_ray = cs["extend_pairs"] / max(1, cs["extend_walkers"])
    prims_per_ray = cs["extend_prims"] / max(1, cs["extend_walkers"])
    spairs_per_ray = cs["shadow_pairs"] / max(1, cs["shadow_walkers"])
    sprims_per_ray = cs["shadow_prims"] / max(1, cs["shadow_walkers"])
    dev.set_profiling(False, False)

    for i in range(a.warmup):
        step(i)
    dev.flush()  # the late paths of the warm-up batches finish (and are folded into the accumulator) before it is cleared
    barrier()
    acc.zero_()
    dev.set_profiling(True, False)
    dev.reset_stats()
    barrier()
    t0 = time.perf_counter()
    def deliver(n_spp):
        if comm is not None:
            dev.reduce_accumulator(comm, 0)  # inside the boundary: finishes the paths in flight, then one ncclReduce on the handle's stream
        else:
            dev.flush()  # the tails of the last batches run on the library's own streams: order the reduce after them
            D.reduce_accumulator(acc, dst=0)  # RCCL over xGMI: 16*W*H bytes, once (no-op for one rank)
        return dev.finalize(n_spp) if rank == 0 else None

    if strong:
        # configs[3]: every step is a whole frame of total_spp samples -- cleared accumulator, this rank's share of the samples,
        # the last paths finished, the reduce, RGB8 on rank 0 -- so the drain and the reduce are paid per frame, as a caller
        # waiting for that frame would pay them
        for i in range(a.steps):
            acc.zero_()
            step(i)
            pixels = deliver(step_spp)
    else:
        for i in range(a.steps):
            step(i)
        pixels = deliver(a.steps * step_spp)
    barrier()
    dt = time.perf_counter() - t0
    st = dev.stats()
    dev.set_profiling(False, False)

    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    rays = torch.tensor([st["extend_rays"] + st["shadow_rays"], st["extend_rays"], st["shadow_rays"], st["primary_samples"],
                         st["extend_walkers"] + st["shadow_walkers"]], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    dt = float(tt.item())
    total_rays, ext_rays, sh_rays, paths, walked = [float(x) for x in rays.tolist()]

    if rank == 0:
        bytes_per_extend_ray = 64.0 * pairs_per_ray + 48.0 * prims_per_ray + 44.0  # per ray that walks the BVH
        ext_bytes = st["extend_walkers"] * bytes_per_extend_ray
        # time during which the kernel was running: with one lane the sum of its launches; with two lanes two launches, one per lane,
        # run side by side and each one's own duration includes the time it shared the chip with the other, so the bytes of all
        # launches are divided by the UNION of their intervals (rtmi_stats.ms_extend_union; equal to the sum for one lane)
        lanes = dev.get_option("lanes")
        ext_s = (st["ms_extend_union"] if st["ms_extend_union"] > 0 else st["ms_extend"]) * 1e-3
        n_launch = max(1, st["n_extend_launches"])
        achieved = ext_bytes / ext_s / 1e9 if ext_s > 0 else 0.0
        per_launch = ext_bytes / (st["ms_extend"] * 1e-3) / 1e9 if st["ms_extend"] > 0 else 0.0
        peak = 8000.0
        # a bounce's shadow walk shares the chip with the next bounce's closest-hit walk: each launch's
        # event-timed duration includes the other's share.  Both walks over the windows they share (shade to shade):
        any_bytes = st["shadow_walkers"] * (64.0 * spairs_per_ray + 48.0 * sprims_per_ray + 48.0)
        walk_s = (st["ms_walk_union"] if st["ms_walk_union"] > 0 else st["ms_walk"]) * 1e-3  # (union over the lanes, as above)
        pair_achieved = (ext_bytes + any_bytes) / walk_s / 1e9 if walk_s > 0 else None
        cfg = "configs[3]" if strong and (W, H) == (3840, 2160) else "configs[2]"
        if a.zoom != 1.0 or a.tube != 0.075:
            cfg = "NOT a config of BASELINE.json (mesh-dominated view for the walk kernels: zoom %g, tube radius %g)" % (a.zoom, a.tube)
        per = ("one frame of %d spp per step split over the ranks" % a.total_spp) if strong else ("%d spp per step per GPU" % spp)
        out = {
            "metric": "Mrays/s (whole node), path-traced OBJ scene 1080p", "value": total_rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: path-traced torus-knot scene (diffuse, mirror, glass) in the reference room built in %gx units, %d triangles, "
                                   "%dx%d, %s, samples split by index over ranks, one RCCL reduce + RGB8 finalize %s"
                                   % (cfg, scene.scale, info["triangles"], W, H, per, "per step (every step is a delivered frame)" if strong else "at the end"), "scene_scale": scene.scale,
                       "triangles": info["triangles"], "bvh_nodes": info["nodes_used"], "bvh_depth": info["max_depth"], "builder": a.builder,
                       "width": W, "height": H, "spp_per_step": step_spp // v_world, "total_spp": a.steps * step_spp, "parallelism": "sample-split x%d" % world,
                       "reduce": a.reduce, "lanes_per_gpu": lanes, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")},
            "mpaths_per_s": paths / dt / 1e6, "extend_rays": ext_rays, "shadow_rays": sh_rays,
            "extend_walkers": float(st["extend_walkers"]), "shadow_walkers": float(st["shadow_walkers"]), "rays_per_path": total_rays / max(1.0, paths),
            # what the headline counts: every closest-hit and any-hit ray.  Most of them end on the room's analytic primitives inside
            # the kernel that created them; the rays that enter the BVH's root box and walk the tree (rank 0's counters):
            "walked_mrays_per_s": walked / dt / 1e6, "primary_mesh_fraction": primary_mesh_fraction,
            "extend_walk_fraction": float(st["extend_walkers"]) / max(1.0, float(st["extend_rays"])),
            "shadow_walk_fraction": float(st["shadow_walkers"]) / max(1.0, float(st["shadow_rays"])),
            # host time from rtmi_flush to the return of rtmi_finalize: the last paths of the render (+ reduce and finalize)
            "drain_ms": st["ms_drain"],
            "l2_hit_rate": pmc.get("l2_hit_rate"), "l2_hit_rate_source": pmc_file,
            "scene_build_s": build_s,
            "roofline": {"bound": "hbm", "kernel": "k_trace<closest>", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                         "traffic": a.traffic_bytes_per_launch, "traffic_source": traffic_source, "bytes_per_launch": ext_bytes / n_launch, "avg_launch_ms": st["ms_extend"] / n_launch,
                         "lanes": lanes, "kernel_running_ms": ext_s * 1e3, "sum_of_launch_ms": st["ms_extend"],
                         "per_launch": {"note": "bytes per launch / mean launch duration; with two lanes a launch shares the chip with the other lane's launch of the same kernel",
                                        "achieved": per_launch, "frac": per_launch / peak},
                         "launches": st["n_extend_launches"], "bytes_per_walked_ray": bytes_per_extend_ray, "node_pairs_per_walked_ray": pairs_per_ray,
                         "leaf_prims_per_walked_ray": prims_per_ray, "shadow_node_pairs_per_walked_ray": spairs_per_ray,
                         "shadow_leaf_prims_per_walked_ray": sprims_per_ray, "longest_walk_steps": [cs["max_extend_steps"], cs["max_shadow_steps"]],
                         "both_walks": {"note": "k_trace<any> of bounce b runs beside k_trace<closest> of bounce b+1; algorithmic bytes of both / time of the windows they share",
                                        "achieved": pair_achieved, "frac": pair_achieved / peak if pair_achieved else None, "window_ms": walk_s * 1e3, "sum_of_window_ms": st["ms_walk"],
                                        "windows": st["n_walk_windows"]}},
            "kernel_ms": {"extend": st["ms_extend"], "shade": st["ms_shade"], "connect": st["ms_connect"], "other": st["ms_other"], "walk_windows": st["ms_walk"]},
        }
        if share > 1:
            out["rehearsal"] = "rank 0's share of a %d-rank %s-scaling run, rendered by one process: per-rank time without the reduce; not a whole-node figure" % (share, a.scaling)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, W, H, seeds, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def rccl_communicator(dist, rank, world, local_rank):
    """An RCCL communicator of this process's own for rtmi_reduce_accumulator (a C++ host would call ncclCommInitRank the same
    way): rank 0 makes the unique id, torch.distributed carries its 128 bytes to the others."""
    import ctypes as C
    import torch
    lib = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            lib = C.CDLL(name)
            break
        except OSError:
            continue
    if lib is None:
        raise SystemExit("--reduce rtmi: librccl.so.1 not found")
    uid = (C.c_byte * 128)()
    if rank == 0 and lib.ncclGetUniqueId(C.byref(uid)) != 0:
        raise SystemExit("ncclGetUniqueId failed")
    t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device="cuda")
    dist.broadcast(t, 0)
    uid = (C.c_byte * 128).from_buffer_copy(bytes(t.cpu().tolist()))
    comm = C.c_void_p()

    class Uid(C.Structure):
        _fields_ = [("internal", C.c_byte * 128)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    u = Uid()
    C.memmove(C.byref(u), uid, 128)
    if lib.ncclCommInitRank(C.byref(comm), world, u, rank) != 0:
        raise SystemExit("ncclCommInitRank failed")
    return comm.value


def cpu_baseline(scene, W, H, seeds, budget_s):
    """The oracle (CPU port of Renderer::Sample, OpenMP rows like renderer.cpp:561) on the host cores of this box,
    one sample per pixel over every `row_step`-th row, sized to about `budget_s` seconds."""
    from raytracer_amd import api
    from oracle.oracle import Oracle
    share = os.environ.get("OMP_NUM_THREADS")
    orc = Oracle(scene.desc)
    if not share:
        orc.set_threads(min(16, os.cpu_count() or 1))  # the box's CPU share for one GPU
    cores = orc.max_threads()
    cam = api.default_camera(W, H, scene.scale)
    # calibrate on every 16th row (twice: the first call also starts the OpenMP threads), then size the sample to about
    # budget_s: a row subset of one sample when a whole frame is too slow, else several whole samples per pixel
    c = np.zeros(7, np.uint64)
    orc.sample_rows(W, H, cam, seeds, 0, 1, 1, 0, 64, None, c)
    t0 = time.perf_counter()
    orc.sample_rows(W, H, cam, seeds, 0, 1, 1, 0, 16, None, c)
    per_row = (time.perf_counter() - t0) / len(range(0, H, 16))
    frame_s = H * per_row
    row_step = max(1, int(np.ceil(frame_s / budget_s)))
    n_spp = max(1, int(budget_s / frame_s)) if row_step == 1 else 1
    c = np.zeros(7, np.uint64)
    t0 = time.perf_counter()
    orc.sample_rows(W, H, cam, seeds, 0, n_spp, 1, 0, row_step, None, c)
    dt = time.perf_counter() - t0
    rows = len(range(0, H, row_step))
    rays = float(c[0] + c[1])
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": int(cores), "kind": "port",
            "sample": "%d spp over %d of %d rows (every %d-th) of the same %dx%d workload, %.1f s" % (n_spp, rows, H, row_step, W, H, dt),
            "mpaths_per_s": n_spp * rows * W / dt / 1e6, "node_tests_per_extend_ray": float(c[2]) / max(1.0, float(c[0])),
            "prim_tests_per_extend_ray": float(c[3]) / max(1.0, float(c[0]))}


if __name__ == "__main__":
    main()
