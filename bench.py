#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s, path-traced ~1M-triangle OBJ scene at 1920x1080 (BASELINE.json configs[2]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" is one Renderer::Tick-equivalent pass of the hot path over one batch: every rank
path-traces --spp-per-step samples for every pixel (ray generation, extend, shade, connect, ... until every path
has terminated) and adds them into its float4 accumulator in HBM.  Samples are split by index over the ranks
(rank r renders sample indices r, r+N, r+2N, ...), so per-GPU work is fixed as N grows (weak scaling:
N GPUs deliver N x K x spp-per-step samples per pixel).  At the end of the timed region the accumulators are
summed onto rank 0 with ONE RCCL reduce (torch.distributed, backend nccl) and rank 0 converts to RGB8 on device.
The scene is written as Wavefront OBJ files (once, cached) and loaded through Scene::LoadObject (scene.h:222-291);
scene, BVH, camera, seed table and all queues are resident in HBM before the timed region starts.

value = (closest-hit + any-hit rays traversed by all ranks, from device counters) / (max over ranks of the time of
the K steps + reduce + finalize), in Mrays/s.  The reference's own "Mrays/s" (primary samples/s, renderer.cpp:639)
is reported beside it as mpaths_per_s, and walked_mrays_per_s counts only the rays that entered the BVH.

roofline: one entry per bulk kernel (k_shade, k_trace<closest>, k_trace<any>) under roofline.kernels, each with its
launches, summed and mean launch duration (hipEvents recorded on the stream the kernel runs on, inside librtmi:
rtmi_set_profiling), the union of its launch intervals (two lanes' launches of one kernel overlap), its algorithmic
bytes and two fractions of the 8 TB/s HBM peak: algorithmic bytes per launch / mean launch duration,
and counter traffic per launch (committed rocprofv3 --pmc passes, profiles/r03_traffic.json) / mean launch duration.
The top-level roofline fields are those of the kernel with the largest summed duration; `achieved` is its per-launch
algorithmic figure -- bytes per launch / mean launch duration, the quantity a reader can recompute from
profiles/r03_kernel_stats.csv.  The walks' bytes are counted twice: on the reference's node layout (32-byte nodes =
64-byte pairs) and on the tree the timed kernel actually walks (64-byte 4-wide records + 32-byte leaf
boxes fetched to confirm a hit), by counting instantiations of the same kernels over the first step's samples.

mesh_view (N = 1): a second, shorter timed leg of the same scene seen through a narrow field of view (77 % of the
camera rays hit triangles), where most rays walk the BVH: the headline's Mrays/s moves with the camera, the walked
rate does not.

cpu_baseline: the CPU restatement (oracle/, kind "port"; validated bit-exact against the compiled reference) timed
on this box's host cores over a bounded sample of the same workload.  N=1, rank 0 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PEAK_GBS = 8000.0  # MI355X HBM3E, MI355X_MICROARCH.md


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
    ap.add_argument("--scene", choices=["obj", "memory"], default="obj",
                    help="obj (default): the knots are written as OBJ files once (--obj-cache) and loaded through Scene::LoadObject; memory: added as triangle arrays")
    ap.add_argument("--obj-cache", default=os.path.join(os.environ.get("TMPDIR", "/tmp"), "rtmi_bench_obj"))
    ap.add_argument("--zoom", type=float, default=1.0, help="NOT the headline: shrink the screen rectangle about its centre (a view filled by the middle knot)")
    ap.add_argument("--tube", type=float, default=0.075, help="NOT the headline: tube radius of the knots (0.075 in every config)")
    ap.add_argument("--mesh-view-steps", type=int, default=5, help="steps of the second timed leg (zoom 0.15: the mesh-dominated view); 0 = skip.  N = 1, weak scaling only")
    ap.add_argument("--hw-queues", type=int, default=16,
                    help="GPU_MAX_HW_QUEUES for this process (HIP's default is 4; 0 = leave the environment alone).  With 12 or more this "
                         "script asks the library for two lanes (two batches in flight side by side, option lanes)")
    ap.add_argument("--lanes", type=int, default=0, help="option lanes of the library: 0 = 2 when the process has >= 12 hardware queues, else 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the headline): every rank renders --spp-per-step samples per step.  strong: a step is ONE frame of "
                         "--total-spp samples split by index over the ranks (BASELINE.json configs[3]: 256 spp at 3840x2160 over 2/4/8 GPUs)")
    ap.add_argument("--total-spp", type=int, default=256, help="--scaling strong: samples per pixel per step, all ranks together")
    ap.add_argument("--share-of", type=int, default=1,
                    help="single process only: render the share of rank 0 of this many ranks (sample indices 0, N, 2N, ...), i.e. what ONE GPU of an "
                         "N-GPU run does between the start and the reduce; the JSON line is then a per-rank rehearsal, not a whole-node figure")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="--scaling strong: frames rendered side by side on handles of their own, so that one frame's last paths finish beside the next "
                         "frames' bulk (raytracer_amd.distributed.FramePipeline).  0 = 4 (3: +8 % per frame, 5: same, 6: +1 %); 1 = one handle, every frame pays its end alone")
    ap.add_argument("--slots", type=int, default=0, help="option slots of every handle (0: the library's 4; 2 per handle with frames in flight: queue memory)")
    ap.add_argument("--reduce", choices=["torch", "rtmi"], default="torch",
                    help="who sums the accumulators: torch.distributed (default) or rtmi_reduce_accumulator on an RCCL communicator of its own")
    return ap.parse_args()


def committed_pmc():
    """PMC counters cannot be collected inside this run (rocprofv3 --pmc is its own pass per counter group): the figures of the
    committed passes of the same command (tools/profile_round.sh -> tools/traffic.py), newest round first.  Returns (dict, file)."""
    for name in ("r03_traffic.json", "r02_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            return d["kernels"], "profiles/" + name, d.get("command")
        except (OSError, KeyError, ValueError):
            continue
    return {}, None, None


class Leg:
    """One timed region: K steps of one view, reduce + finalize at the end (weak) or per step (strong)."""

    def __init__(self, a, torch, dist, handles, rank, world, use_dist, comm):
        self.a, self.torch, self.dist, self.handles = a, torch, dist, handles
        self.dev, self.stream, self.acc = handles[0]  # weak scaling, counting and the unpipelined strong loop use the first handle only
        self.rank, self.world, self.use_dist, self.comm = rank, world, use_dist, comm
        share = max(1, a.share_of)
        self.v_world, self.v_rank = (share, 0) if share > 1 else (world, rank)  # the split this process renders its part of
        self.strong = a.scaling == "strong"
        # samples per step over all (virtual) ranks: weak = spp-per-step each, strong = one frame of total-spp
        self.step_spp = a.total_spp if self.strong else a.spp_per_step * self.v_world
        # markers for tools/trace_union.py: two fill kernels that nothing else in the process launches bracket the timed region
        self.mark_a = torch.empty(1, dtype=torch.int16, device="cuda")  # (empty: creating them must not launch the marker kernel)
        self.mark_b = torch.empty(1, dtype=torch.int8, device="cuda")
        self.delivered = []  # strong scaling: seconds after the start of the timed region at which each frame was in the caller's hands

    def step(self, i, dev=None):
        from raytracer_amd import api, distributed as D
        dev = dev or self.dev
        # step i covers the global sample indices [i*step_spp, (i+1)*step_spp); this rank renders every v_world-th one
        D.render_split(lambda b, n, st: dev.render(api.PATHTRACING, b, n, st), i * self.step_spp, self.step_spp, self.v_rank, self.v_world)

    def barrier(self):
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def deliver(self, n_spp, handle=None, ordered=None):
        from raytracer_amd import distributed as D
        import contextlib
        dev, stream, acc = handle or self.handles[0]
        dev.flush()  # the tails of the last batches run on the library's own streams: the reduce is ordered after them
        with ordered or contextlib.nullcontext():  # frames in flight: the frames' collectives are issued in frame order on every rank
            if self.comm is not None:
                # inside the boundary: one ncclReduce on the handle's stream (one communicator, a stream per handle: the library makes
                # each reduce on a communicator execute after the one before it)
                dev.reduce_accumulator(self.comm, 0)
            else:
                D.reduce_accumulator(acc, dst=0)  # RCCL over xGMI: 16*W*H bytes, once (no-op for one rank)
        return dev.finalize(n_spp) if self.rank == 0 else None

    def frame(self, k, i, ordered):
        """one delivered frame of the strong-scaling loop on handle k: cleared accumulator, this rank's share of the frame's samples,
        the last paths finished, the reduce, RGB8 on rank 0 -- the calls of the unpipelined loop, on a handle of the frame's own"""
        dev, stream, acc = self.handles[k]
        with self.torch.cuda.stream(stream):
            acc.zero_()
            self.step(i, dev)
            return self.deliver(self.step_spp, self.handles[k], ordered)

    def count(self):
        """not timed: records per walked ray on this view, on the reference's pairs (1) and on the 4-wide nodes the timed kernels walk (2)"""
        out = {}
        for mode in (1, 2):
            self.dev.set_profiling(False, mode)
            self.dev.reset_stats()
            self.step(0)
            self.dev.sync()
            out[mode] = self.dev.stats()
        self.dev.set_profiling(False, 0)
        return out

    def run(self, steps, warmup):
        from raytracer_amd import distributed as D
        dev, acc = self.dev, self.acc
        used = self.handles if self.strong else self.handles[:1]
        for i in range(warmup):
            for d, _, _ in used:
                self.step(i, d)
        for d, _, _ in used:
            d.flush()  # the late paths of the warm-up batches finish (and are folded into the accumulator) before it is cleared
        self.barrier()
        for d, _, a_ in used:
            a_.zero_()
            d.set_profiling(True, 0)
            d.reset_stats()
        self.barrier()
        self.mark_a.fill_(1)
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        if self.strong:
            # configs[3]: every step is a whole frame of total_spp samples -- cleared accumulator, this rank's share of the samples,
            # the last paths finished, the reduce, RGB8 on rank 0 -- so the drain and the reduce are paid per frame, as a caller
            # waiting for that frame would pay them
            # as a caller waiting for that frame would pay them; with frames in flight the following frames' bulk runs meanwhile
            if len(self.handles) > 1:
                pipe = D.FramePipeline(len(self.handles))
                for f in [pipe.submit(i, self.frame) for i in range(steps)]:
                    f.result()
                    self.delivered.append(time.perf_counter() - t0)
                pipe.close()
            else:
                for i in range(steps):
                    acc.zero_()
                    self.step(i)
                    self.deliver(self.step_spp)
                    self.delivered.append(time.perf_counter() - t0)
        else:
            for i in range(steps):
                self.step(i)
            self.deliver(steps * self.step_spp)
        self.barrier()
        dt = time.perf_counter() - t0
        self.mark_b.fill_(1)
        self.torch.cuda.synchronize()
        st = dev.stats()
        dev.set_profiling(False, 0)
        for d, _, _ in used[1:]:  # frames in flight: the handles' counters and timings add up
            o = d.stats()
            d.set_profiling(False, 0)
            for key, v in o.items():
                if isinstance(v, (int, float)) and not key.startswith("max_"):
                    st[key] = st[key] + v
                elif key.startswith("max_"):
                    st[key] = max(st[key], v)
        torch, dist = self.torch, self.dist
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        rays = torch.tensor([st["extend_rays"] + st["shadow_rays"], st["extend_rays"], st["shadow_rays"], st["primary_samples"],
                             st["extend_walkers"] + st["shadow_walkers"]], dtype=torch.float64, device="cuda")
        if self.use_dist:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        return float(tt.item()), [float(x) for x in rays.tolist()], st


def kernel_rooflines(st, cnt, pmc):
    """The three bulk kernels of the timed region (rank 0's counters and event timings).  Algorithmic bytes:
      k_trace<closest>  per walked ray 64 B x pair records + 48 B x leaf primitive records + 44 B (32 B ray read, 4 B walk index, 8 B hit write)
      k_trace<any>      per walked ray 64 B x pairs + 48 B x leaf records + 48 B (the queue entry)
      k_shade           per vertex 52 B state read + 16 B hit attributes; 52 B per survivor written; 4 B per walker queued; 48 B per
                        shadow walker queued; 32 B per contribution added to E[slot] (upper bound: every shadow ray that did not walk)
    and for the walks the same on the tree the kernel walks: 64 B x 4-wide records + 64 B x pairs (rays the 4-wide nodes cannot take) +
    48 B x leaf records + 32 B x leaf boxes fetched to confirm a hit."""
    c1, c2 = cnt[1], cnt[2]
    ew, sw = max(1, c1["extend_walkers"]), max(1, c1["shadow_walkers"])
    per = {
        "closest": {"ref": 64.0 * c1["extend_pairs"] / ew + 48.0 * c1["extend_prims"] / ew + 44.0,
                    "walked": (64.0 * (c2["extend_wide_nodes"] + c2["extend_pairs"]) + 48.0 * c2["extend_prims"] + 32.0 * c2["extend_confirms"]) / max(1, c2["extend_walkers"]) + 44.0,
                    "pairs": c1["extend_pairs"] / ew, "prims": c1["extend_prims"] / ew, "wide_nodes": c2["extend_wide_nodes"] / max(1, c2["extend_walkers"]),
                    "confirms": c2["extend_confirms"] / max(1, c2["extend_walkers"])},
        "any": {"ref": 64.0 * c1["shadow_pairs"] / sw + 48.0 * c1["shadow_prims"] / sw + 48.0,
                "walked": (64.0 * (c2["shadow_wide_nodes"] + c2["shadow_pairs"]) + 48.0 * c2["shadow_prims"] + 32.0 * c2["shadow_confirms"]) / max(1, c2["shadow_walkers"]) + 48.0,
                "pairs": c1["shadow_pairs"] / sw, "prims": c1["shadow_prims"] / sw, "wide_nodes": c2["shadow_wide_nodes"] / max(1, c2["shadow_walkers"]),
                "confirms": c2["shadow_confirms"] / max(1, c2["shadow_walkers"])},
    }
    V, P, S = float(st["extend_rays"]), float(st["primary_samples"]), float(st["shadow_rays"])
    We, Ws = float(st["extend_walkers"]), float(st["shadow_walkers"])
    shade_bytes = 68.0 * V + 52.0 * max(0.0, V - P) + 4.0 * We + 48.0 * Ws + 32.0 * max(0.0, S - Ws)
    rows = {
        "k_shade": dict(bytes=shade_bytes, sum_ms=st["ms_shade"], union_ms=st["ms_shade_union"], launches=st["n_shade_launches"], units=V, unit="vertices"),
        "k_trace<closest>": dict(bytes=We * per["closest"]["ref"], bytes_walked_tree=We * per["closest"]["walked"], sum_ms=st["ms_extend"], union_ms=st["ms_extend_union"],
                                 launches=st["n_extend_launches"], units=We, unit="walked rays", per_unit=per["closest"]),
        "k_trace<any>": dict(bytes=Ws * per["any"]["ref"], bytes_walked_tree=Ws * per["any"]["walked"], sum_ms=st["ms_connect"], union_ms=st["ms_connect_union"],
                             launches=st["n_connect_launches"], units=Ws, unit="walked rays", per_unit=per["any"]),
    }
    out = {}
    for name, r in rows.items():
        n = max(1, r["launches"])
        avg_ms = r["sum_ms"] / n
        e = {"launches": r["launches"], "sum_of_launch_ms": r["sum_ms"], "avg_launch_ms": avg_ms, "union_ms": r["union_ms"] or r["sum_ms"],
             "algorithmic_bytes_per_launch": r["bytes"] / n, "bytes_per_unit": r["bytes"] / max(1.0, r["units"]), "unit": r["unit"]}
        e["achieved_per_launch"] = r["bytes"] / (r["sum_ms"] * 1e-3) / 1e9 if r["sum_ms"] > 0 else 0.0  # = bytes per launch / mean launch duration
        e["frac_per_launch"] = e["achieved_per_launch"] / PEAK_GBS
        e["achieved_over_union"] = r["bytes"] / (e["union_ms"] * 1e-3) / 1e9 if e["union_ms"] > 0 else 0.0
        e["frac_over_union"] = e["achieved_over_union"] / PEAK_GBS
        if "bytes_walked_tree" in r:
            e["walked_tree_bytes_per_launch"] = r["bytes_walked_tree"] / n
            e["walked_tree_frac_per_launch"] = r["bytes_walked_tree"] / (r["sum_ms"] * 1e-3) / 1e9 / PEAK_GBS if r["sum_ms"] > 0 else 0.0
            e["per_walked_ray"] = r["per_unit"]
        k = pmc.get(name)
        if k and "traffic_bytes_per_launch" in k:
            e["traffic_bytes_per_launch"] = k["traffic_bytes_per_launch"]
            if k.get("avg_launch_ms"):  # the traffic passes' own launch time (counter collection serialises the kernels)
                e["traffic_rate_in_pmc_pass"] = k["traffic_bytes_per_launch"] / (k["avg_launch_ms"] * 1e-3) / 1e9
            e["traffic_frac_per_launch"] = k["traffic_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / PEAK_GBS if avg_ms > 0 else None
            e["l2_hit_rate"] = k.get("l2_hit_rate")
        out[name] = e
    return out


def main():
    a = parse()
    # hardware queues: the library's streams (per lane: main, side, four tails) must not share them; the HIP runtime reads this
    # when it starts, i.e. at the first import of torch below
    if a.hw_queues > 0 and int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < a.hw_queues:
        os.environ["GPU_MAX_HW_QUEUES"] = str(min(a.hw_queues, 32))
    lanes = a.lanes if a.lanes > 0 else (2 if int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) >= 12 else 1)
    pmc, pmc_file, pmc_cmd = committed_pmc()
    import torch
    import torch.distributed as dist
    from raytracer_amd import api

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
    builder = api.BINNED_SAH if a.builder == "binned" else api.REFERENCE_SAH
    t0 = time.time()
    if a.scene == "obj":
        scene, ingest = api.mesh_scene_obj(a.tris, a.obj_cache, builder, a.tube)
    else:
        scene, ingest = api.mesh_scene(a.tris, builder, tube=a.tube), None
    info = scene.info()
    build_s = time.time() - t0
    strong = a.scaling == "strong"
    in_flight = (a.frames_in_flight or 4) if strong else 1
    if in_flight > 1 and a.lanes == 0:
        lanes = 1  # the handles' frames already run side by side
    slots = a.slots or (2 if in_flight > 1 else 0)
    n_analytic = info["quads"] + info["spheres"] + info["cubes"] + info["planes"]
    seeds = api.seed_table(W * H)
    # Everything a handle does runs on ONE explicit torch stream: the library enqueues on it (rtmi_set_stream takes the stream as it
    # is), rtmi_flush makes it wait for the library's internal streams, and the collective -- issued while it is torch's current
    # stream -- is ordered after it by torch's own event.
    handles = []
    for k in range(in_flight):
        d = api.Device(W, H, local_rank, lanes=lanes)
        if slots:
            d.set_option("slots", slots)
        d.upload(scene.desc)
        d.set_seeds(seeds)
        stream = torch.cuda.Stream()
        acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        d.set_stream(stream.cuda_stream)
        d.set_accumulator(acc.data_ptr())
        handles.append((d, stream, acc))
    dev, stream, acc = handles[0]
    torch.cuda.set_stream(stream)
    if max(1, a.share_of) > 1 and world > 1:
        raise SystemExit("--share-of is a single-process rehearsal")
    comm = rccl_communicator(dist, rank, world, local_rank) if a.reduce == "rtmi" and use_dist else None
    leg = Leg(a, torch, dist, handles, rank, world, use_dist, comm)

    def view(zoom, steps, warmup):
        for d, _, _ in handles:
            d.set_camera(api.zoom_camera(api.default_camera(W, H, scene.scale), zoom))
        # how much of the view is mesh: camera rays whose nearest hit is a triangle (the analytic primitives come first in the numbering)
        mesh_fraction = float((dev.trace_primary()[0] >= n_analytic).mean())
        cnt = leg.count()
        dt, rays, st = leg.run(steps, warmup)
        return dt, rays, st, cnt, mesh_fraction

    dt, (total_rays, ext_rays, sh_rays, paths, walked), st, cnt, primary_mesh_fraction = view(a.zoom, a.steps, a.warmup)
    mesh_view = None
    if world == 1 and not strong and a.mesh_view_steps > 0 and a.zoom == 1.0 and max(1, a.share_of) == 1:
        mdt, (mt, me, ms, mp, mw), mst, mcnt, mfrac = view(0.15, a.mesh_view_steps, 1)
        mk = kernel_rooflines(mst, mcnt, {})
        mesh_view = {"note": "second timed leg of this run, NOT the headline: the same scene and spp per step through a narrower field of view (zoom 0.15: the middle, "
                             "mirror knot fills the view), where most rays walk the BVH", "steps": a.mesh_view_steps, "warmup": 1,
                     "value": mt / mdt / 1e6, "unit": "Mrays/s", "ms_per_step": mdt * 1e3 / a.mesh_view_steps, "walked_mrays_per_s": mw / mdt / 1e6,
                     "mpaths_per_s": mp / mdt / 1e6, "primary_mesh_fraction": mfrac, "extend_walk_fraction": float(mst["extend_walkers"]) / max(1.0, me),
                     "shadow_walk_fraction": float(mst["shadow_walkers"]) / max(1.0, ms), "drain_ms": mst["ms_drain"], "dropped_paths": mst["dropped_paths"],
                     "roofline": {"bound": "hbm", "peak": PEAK_GBS, "unit": "GB/s", "kernels": mk}}

    if rank == 0:
        kernels = kernel_rooflines(st, cnt, pmc)
        top = max(kernels, key=lambda k: kernels[k]["sum_of_launch_ms"])
        tk = kernels[top]
        cfg = "configs[3]" if strong and (W, H) == (3840, 2160) else "configs[2]"
        if a.zoom != 1.0 or a.tube != 0.075:
            cfg = "NOT a config of BASELINE.json (mesh-dominated view for the walk kernels: zoom %g, tube radius %g)" % (a.zoom, a.tube)
        per = ("one frame of %d spp per step split over the ranks" % a.total_spp) if strong else ("%d spp per step per GPU" % a.spp_per_step)
        out = {
            "metric": "Mrays/s (whole node), path-traced OBJ scene 1080p", "value": total_rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: path-traced torus-knot scene (diffuse, mirror, glass; %s) in the reference room built in %gx units, %d triangles, "
                                   "%dx%d, %s, samples split by index over ranks, one RCCL reduce + RGB8 finalize %s"
                                   % (cfg, "three OBJ files through Scene::LoadObject" if a.scene == "obj" else "triangle arrays", scene.scale, info["triangles"], W, H, per,
                                      "per step (every step is a delivered frame)" if strong else "at the end"), "scene_scale": scene.scale,
                       "triangles": info["triangles"], "bvh_nodes": info["nodes_used"], "bvh_depth": info["max_depth"], "builder": a.builder, "scene_source": a.scene,
                       "width": W, "height": H, "spp_per_step": leg.step_spp // leg.v_world, "total_spp": a.steps * leg.step_spp, "parallelism": "sample-split x%d" % world,
                       "reduce": a.reduce, "lanes_per_gpu": dev.get_option("lanes"), "frames_in_flight": in_flight, "slots": dev.get_option("slots"), "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")},
            "mpaths_per_s": paths / dt / 1e6, "extend_rays": ext_rays, "shadow_rays": sh_rays,
            "extend_walkers": float(st["extend_walkers"]), "shadow_walkers": float(st["shadow_walkers"]), "rays_per_path": total_rays / max(1.0, paths),
            # what the headline counts: every closest-hit and any-hit ray.  Most of them end on the room's analytic primitives inside
            # the kernel that created them; the rays that enter the BVH's root box and walk the tree (rank 0's counters):
            "walked_mrays_per_s": walked / dt / 1e6, "primary_mesh_fraction": primary_mesh_fraction,
            "extend_walk_fraction": float(st["extend_walkers"]) / max(1.0, float(st["extend_rays"])),
            "shadow_walk_fraction": float(st["shadow_walkers"]) / max(1.0, float(st["shadow_rays"])),
            # host time from rtmi_flush to the return of rtmi_finalize: the last paths of the render (+ reduce and finalize)
            "drain_ms": st["ms_drain"], "dropped_paths": st["dropped_paths"],
            "scene_build_s": build_s, "obj_ingest_s": ingest["obj_ingest_s"] if ingest else None, "obj": ingest,
            "roofline": {"bound": "hbm", "kernel": top, "achieved": tk["achieved_per_launch"], "peak": PEAK_GBS, "unit": "GB/s", "frac": tk["frac_per_launch"],
                         "traffic": tk.get("traffic_bytes_per_launch"),
                         "traffic_source": ("%s: separate rocprofv3 --pmc passes of an earlier run of this command (%s), not this run" % (pmc_file, pmc_cmd)) if pmc_file else None,
                         "note": "achieved = algorithmic bytes per launch / mean launch duration of the kernel with the largest summed duration; every bulk kernel under `kernels`",
                         "bytes_per_launch": tk["algorithmic_bytes_per_launch"], "avg_launch_ms": tk["avg_launch_ms"], "launches": tk["launches"],
                         "lanes": dev.get_option("lanes"), "kernels": kernels,
                         "longest_walk_steps": [cnt[1]["max_extend_steps"], cnt[1]["max_shadow_steps"]],
                         "both_walks": {"note": "k_trace<any> of bounce b runs beside k_trace<closest> of bounce b+1: algorithmic bytes of both / union of the windows they share",
                                        "window_union_ms": st["ms_walk_union"] or st["ms_walk"], "sum_of_window_ms": st["ms_walk"], "windows": st["n_walk_windows"]}},
            "kernel_ms": {"extend": st["ms_extend"], "shade": st["ms_shade"], "connect": st["ms_connect"], "other": st["ms_other"], "walk_windows": st["ms_walk"]},
        }
        bw = out["roofline"]["both_walks"]
        if bw["window_union_ms"] > 0:
            both = (kernels["k_trace<closest>"]["algorithmic_bytes_per_launch"] * kernels["k_trace<closest>"]["launches"]
                    + kernels["k_trace<any>"]["algorithmic_bytes_per_launch"] * kernels["k_trace<any>"]["launches"])
            bw["achieved"] = both / (bw["window_union_ms"] * 1e-3) / 1e9
            bw["frac"] = bw["achieved"] / PEAK_GBS
        if mesh_view:
            out["mesh_view"] = mesh_view
        if strong and len(leg.delivered) >= 4:
            out["frames"] = {"in_flight": in_flight, "delivered_at_ms": [round(t * 1e3, 1) for t in leg.delivered],
                             "steady_state_ms_per_frame": (leg.delivered[-1] - leg.delivered[in_flight - 1]) * 1e3 / (len(leg.delivered) - in_flight),
                             "note": "ms_per_step = the whole timed region / frames (the pipeline's fill and the last frame's end included); steady state = "
                                     "mean interval between deliveries once every handle has delivered a frame"}
        if max(1, a.share_of) > 1:
            out["rehearsal"] = "rank 0's share of a %d-rank %s-scaling run, rendered by one process: per-rank time without the reduce; not a whole-node figure" % (a.share_of, a.scaling)
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
    """The oracle (CPU port of Renderer::Sample, OpenMP rows like renderer.cpp:561) on the host cores of this box: whole
    samples per pixel of the same workload, as many as fit `budget_s` seconds (a row subset of one sample when a frame is slower)."""
    from raytracer_amd import api
    from oracle.oracle import Oracle
    share = os.environ.get("OMP_NUM_THREADS")
    orc = Oracle(scene.desc)
    if not share:
        orc.set_threads(min(16, os.cpu_count() or 1))  # the box's CPU share for one GPU
    cores = orc.max_threads()
    cam = api.default_camera(W, H, scene.scale)
    # calibrate: a strided pass to start the OpenMP threads and warm the caches, then one whole sample, timed
    c = np.zeros(7, np.uint64)
    orc.sample_rows(W, H, cam, seeds, 0, 1, 1, 0, 16, None, c)
    t0 = time.perf_counter()
    orc.sample_rows(W, H, cam, seeds, 0, 1, 1, 0, 4, None, c)
    frame_s = 4.0 * (time.perf_counter() - t0)
    row_step = max(1, int(np.ceil(frame_s / budget_s)))
    n_spp = max(1, int(round(budget_s / frame_s))) if row_step == 1 else 1
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
