"""Multi-rank path on CPU: world_size-2 gloo processes run the product's sample split + reduce
(raytracer_amd/distributed.py) with the oracle standing in for the GPU renderer, and rank 0's reduced
accumulator must equal the single-process image up to float summation order."""
import os
import sys

import numpy as np
import pytest

from raytracer_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("first,count", [(0, 16), (5, 7), (0, 1), (3, 0)])
def test_split_covers_each_sample_once(world, first, count):
    seen = []
    for r in range(world):
        b, n, s = D.split_samples(first, count, r, world)
        assert s == world and n >= 0
        seen += [b + k * s for k in range(n)]
    assert sorted(seen) == list(range(first, first + count))


def test_split_rejects_bad_arguments():
    for args in ((0, 4, 2, 2), (0, 4, -1, 2), (0, -1, 0, 2), (-1, 4, 0, 2), (0, 4, 0, 0)):
        with pytest.raises(ValueError):
            D.split_samples(*args)


W, H, SPP = 64, 36, 6


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from raytracer_amd import api
    from oracle.oracle import Oracle
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = api.HostScene.default()
    orc = Oracle(scene.desc)
    orc.set_threads(2)
    cam, seeds = api.default_camera(W, H), api.seed_table(W * H)
    acc = torch.zeros((H, W, 4), dtype=torch.float32)

    def render(begin, count, stride):  # stand-in for Device.render: accumulates into the tensor's memory
        orc.sample(W, H, cam, seeds, begin, count, stride, acc.numpy())

    mine = D.render_split(render, 0, SPP)
    assert mine == SPP // world
    D.reduce_accumulator(acc, dst=0)
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    import torch.multiprocessing as mp
    from raytracer_amd import api
    from oracle.oracle import Oracle
    out = str(tmp_path / "acc.npy")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    scene = api.HostScene.default()
    ref = Oracle(scene.desc).sample(W, H, api.default_camera(W, H), api.seed_table(W * H), 0, SPP)
    assert np.abs(got[..., :3]).max() > 0
    # same samples, summed in a different order: relative 1e-6, far inside the RMSE 1e-3 bar
    assert np.allclose(got, ref, rtol=2e-6, atol=1e-6)
    rmse = float(np.sqrt(np.mean((got[..., :3].astype(np.float64) - ref[..., :3]) ** 2))) / SPP
    assert rmse <= 1e-6


def test_frame_pipeline_orders_the_collectives_and_the_results():
    """FramePipeline: frame i runs on handle i % n, a handle's frames one after the other, the `ordered` sections strictly in frame
    order whatever the threads' timing, results in submission order; a job that skips its section or raises still passes the turn on"""
    import threading
    import time
    log, lock, running = [], threading.Lock(), {}

    def job(k, i, ordered):
        assert k == i % 3
        with lock:
            assert not running.get(k), "two frames on one handle at once"
            running[k] = True
        time.sleep(0.02 * ((7 * i) % 3))  # frames reach their collective out of order
        if i == 4:
            with lock:
                running[k] = False
            raise RuntimeError("frame 4 fails before its collective")
        if i != 2:  # frame 2 has no collective
            with ordered:
                log.append(i)
                time.sleep(0.005)
        with lock:
            running[k] = False
        return i * i

    pipe = D.FramePipeline(3)
    futures = [pipe.submit(i, job) for i in range(9)]
    with pytest.raises(ValueError):
        pipe.submit(11, job)
    res = []
    for i, f in enumerate(futures):
        if i == 4:
            with pytest.raises(RuntimeError):
                f.result(timeout=30)
            res.append(None)
        else:
            res.append(f.result(timeout=30))
    pipe.close()
    assert res == [0, 1, 4, 9, None, 25, 36, 49, 64]
    assert log == [0, 1, 3, 5, 6, 7, 8]


def test_cpp_frame_pipeline_orders_the_collectives():
    """raytracer_amd/host/frames.h (what a C++ host uses): the same properties through rth_frames_selftest, no GPU"""
    import ctypes as C
    from raytracer_amd import api
    L = api.host_lib()
    order, n = (C.c_int * 64)(), C.c_int()
    assert L.rth_frames_selftest(3, 11, 0b100100, 7, order, C.byref(n)) == 7  # frames 2 and 5 have no collective, frame 7 fails before its own
    assert list(order[:n.value]) == [0, 1, 3, 4, 6, 8, 9, 10]
    assert L.rth_frames_selftest(1, 5, 0, -1, order, C.byref(n)) == 0 and list(order[:n.value]) == [0, 1, 2, 3, 4]
    assert L.rth_frames_selftest(4, 0, 0, -1, order, C.byref(n)) == 0 and n.value == 0
