"""The N>1 path with the real renderer: two processes on the one GPU of the box, each with its own rtmi context,
render their share of the samples (raytracer_amd/distributed.py: rank r renders samples r, r+2, ...), the
accumulators are summed onto rank 0 (gloo here: RCCL wants one GPU per rank; bench.py uses backend "nccl" on the
8-GPU node) and must equal the single-process image up to float summation order, and the oracle's to the RMSE bar."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

W, H, SPP = 320, 180, 6


def _worker(rank, world, port, out_path, backend="gloo"):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from raytracer_amd import api
    from raytracer_amd import distributed as D
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    gpu = rank if backend == "nccl" else 0  # RCCL: one GPU per rank; gloo: both ranks share the box's one GPU
    torch.cuda.set_device(gpu)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = api.HostScene.default()
    d = api.Device(W, H, gpu)
    d.upload(scene.desc)
    d.set_seeds(api.seed_table(W * H))
    d.set_camera(api.default_camera(W, H))
    # as bench.py does it: a caller-owned device accumulator on an explicit torch stream, rtmi_flush, then the consumer on that
    # stream -- no rtmi_read_accumulator in between, whose own synchronisation would hide a missing ordering
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    d.set_stream(stream.cuda_stream)
    d.set_accumulator(acc.data_ptr())
    mine = D.render_split(lambda b, n, st: d.render(api.PATHTRACING, b, n, st), 0, SPP)
    assert mine == SPP // world
    d.flush()
    # what a collective on this stream would read: a copy issued by torch on the caller's stream right after rtmi_flush (gloo
    # reduces host tensors; RCCL wants one GPU per rank and this box has one)
    if backend == "nccl":
        D.reduce_accumulator(acc, dst=0)  # the device tensor itself, on the stream rtmi_flush has just ordered: bench.py's path
        host = acc.cpu()
    else:
        host = acc.cpu()
        D.reduce_accumulator(host, dst=0)
    if rank == 0:
        np.save(out_path, host.numpy())
    dist.barrier()
    d.close()
    dist.destroy_process_group()


def test_two_gpu_ranks_equal_one(tmp_path):
    import torch.multiprocessing as mp
    from raytracer_amd import api
    out = str(tmp_path / "acc.npy")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    scene = api.HostScene.default()  # (keep it alive: desc points into it)
    d = api.Device(W, H, 0)
    d.upload(scene.desc)
    d.set_seeds(api.seed_table(W * H))
    d.render(api.PATHTRACING, 0, SPP, 1)
    one = d.accumulator()
    d.close()
    assert np.abs(one[..., :3]).max() > 0
    assert np.allclose(got, one, rtol=2e-6, atol=1e-6)
    # and against the oracle on the same samples
    from oracle.oracle import Oracle
    ref = Oracle(scene.desc).sample(W, H, api.default_camera(W, H), api.seed_table(W * H), 0, SPP)
    rmse = float(np.sqrt(np.mean((got[..., :3].astype(np.float64) - ref[..., :3]) ** 2))) / SPP
    assert rmse <= 1e-5  # bar 1e-3


def test_two_ranks_over_rccl_when_the_box_has_two_gpus(tmp_path):
    """backend "nccl" (= RCCL), one GPU per rank, the device accumulators reduced directly after rtmi_flush -- what bench.py does
    on the multi-GPU node.  Needs two GPUs: skipped on the one-GPU box."""
    import torch
    import torch.multiprocessing as mp
    from raytracer_amd import api
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU: RCCL wants a GPU per rank (the one-rank RCCL path is test_reduce_accumulator_inside_the_boundary)")
    out = str(tmp_path / "acc_nccl.npy")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out, "nccl"), nprocs=2, join=True)
    got = np.load(out)
    scene = api.HostScene.default()
    d = api.Device(W, H, 0)
    d.upload(scene.desc)
    d.set_seeds(api.seed_table(W * H))
    d.render(api.PATHTRACING, 0, SPP, 1)
    one = d.accumulator()
    d.close()
    assert np.allclose(got, one, rtol=2e-6, atol=1e-6)


def _load_rccl():
    import ctypes as C
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    return None


def _reduce_worker(_, out_path):
    """In a fresh process that imports torch FIRST, as bench.py does: this image has two ROCm runtimes (torch's bundled one
    and /opt/rocm's), and RCCL only works when the whole process uses one of them -- the one loaded first."""
    import ctypes as C
    import torch
    sys.path.insert(0, ROOT)
    from raytracer_amd import api
    lib = _load_rccl()
    if lib is None:
        np.save(out_path, np.zeros(1))
        return
    ndev = min(2, torch.cuda.device_count())
    comms = (C.c_void_p * ndev)()
    devs = (C.c_int * ndev)(*range(ndev))
    assert lib.ncclCommInitAll(comms, ndev, devs) == 0
    scene = api.HostScene.default()
    seeds = api.seed_table(W * H)
    ds = []
    for k in range(ndev):
        d = api.Device(W, H, k)
        d.upload(scene.desc)
        d.set_seeds(seeds)
        d.render(api.PATHTRACING, k, SPP // ndev, ndev)
        ds.append(d)
    for root in (0, -1):  # ncclReduce onto rank 0; with one rank also ncclAllReduce (either must leave the one image as it is)
        if root == -1 and ndev > 1:
            break
        assert lib.ncclGroupStart() == 0
        for k, d in enumerate(ds):
            d.reduce_accumulator(comms[k], root)
        assert lib.ncclGroupEnd() == 0
    got = ds[0].accumulator()
    one = api.Device(W, H, 0)
    one.upload(scene.desc)
    one.set_seeds(seeds)
    one.render(api.PATHTRACING, 0, SPP, 1)
    want = one.accumulator()
    assert api.device_lib().rtmi_reduce_accumulator(ds[0].h, None, 0) == -1
    # frames in flight with the reduce inside (host/frames.h through rth_render_frames): three handles of GPU 0 share rank 0's
    # communicator, every frame's reduce is issued in frame order and chained to the one before; with one rank each frame must come
    # out as the frame of an unpipelined loop without any reduce
    if ndev == 1:
        FR, T = 5, 4
        hs = []
        for _ in range(3):
            h = api.Device(W, H, 0, batch=1, carry_min=0, slots=2)
            h.upload(scene.desc)
            h.set_seeds(seeds)
            hs.append(h)
        px = api.render_frames(hs, FR, T, 0, 1, comms[0])
        for f in range(FR):
            one.reset()
            one.render(api.PATHTRACING, f * T, T, 1)
            assert np.array_equal(px[f], one.finalize(T)), "frame %d" % f
        for h in hs:
            h.close()
    for d in ds + [one]:
        d.close()
    for k in range(ndev):
        lib.ncclCommDestroy(C.c_void_p(comms[k]))
    np.save(out_path, np.stack([got, want]))
    with open(out_path + ".ndev", "w") as f:
        f.write(str(ndev))


def test_reduce_accumulator_inside_the_boundary(tmp_path):
    """rtmi_reduce_accumulator: the sum over ranks happens inside the C ABI, on an RCCL communicator the host made.  One
    process drives every GPU of the box (ncclCommInitAll + group calls): 1 rank here on the one-GPU box (a reduce that must
    leave the image as it is), 2 ranks where a second GPU is visible (sample split, sum == single render)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "reduced.npy")
    mp.spawn(_reduce_worker, args=(out,), nprocs=1, join=True)
    res = np.load(out)
    if res.ndim == 1:
        pytest.skip("no librccl.so.1 on this machine")
    got, want = res
    assert np.abs(want[..., :3]).max() > 0
    if open(out + ".ndev").read() == "1":
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    else:
        assert np.allclose(got, want, rtol=2e-6, atol=1e-6)
