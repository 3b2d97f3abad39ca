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


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from raytracer_amd import api
    from raytracer_amd import distributed as D
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = api.HostScene.default()
    d = api.Device(W, H, 0)
    d.upload(scene.desc)
    d.set_seeds(api.seed_table(W * H))
    d.set_camera(api.default_camera(W, H))
    mine = D.render_split(lambda b, n, st: d.render(api.PATHTRACING, b, n, st), 0, SPP)
    assert mine == SPP // world
    acc = torch.from_numpy(d.accumulator().copy())  # read-back completes the paths in flight
    D.reduce_accumulator(acc, dst=0)
    if rank == 0:
        np.save(out_path, acc.numpy())
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
