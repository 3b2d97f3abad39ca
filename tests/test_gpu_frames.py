"""Strong scaling's frame loop (BASELINE.json configs[3]): frames in flight on handles of their own
(raytracer_amd.distributed.FramePipeline) deliver, frame for frame, the bits of the unpipelined loop -- cleared accumulator, this rank's
share of the frame's samples, the last paths finished, RGB8 -- while the end of one frame runs beside the bulk of the next ones."""
import time

import numpy as np
import pytest

from raytracer_amd import api, distributed as D
from conftest import bits

pytestmark = pytest.mark.gpu


def test_frames_in_flight_equal_the_unpipelined_frames(default_scene):
    W, H, TOTAL, WORLD, FRAMES, IN_FLIGHT = 640, 360, 32, 4, 7, 3
    seeds, cam = api.seed_table(W * H), api.default_camera(W, H)
    # batches of 2 samples that hand their late paths on (carry_min = 0): every frame is four batches and a camera-less one, as at full size

    def handle():
        d = api.Device(W, H, batch=2, carry_min=0, slots=2)
        d.upload(default_scene.desc)
        d.set_seeds(seeds)
        d.set_camera(cam)
        return d

    def frame(d, i):  # rank 0's share of frame i: global samples i*TOTAL + 0, WORLD, 2*WORLD, ...
        d.reset()
        assert D.render_split(lambda b, n, st: d.render(api.PATHTRACING, b, n, st), i * TOTAL, TOTAL, 0, WORLD) == TOTAL // WORLD
        d.flush()
        px = d.finalize(TOTAL // WORLD)
        return px, d.accumulator()

    one = handle()
    want = [frame(one, i) for i in range(FRAMES)]
    one.close()
    assert not np.array_equal(want[0][0], want[1][0])  # frames differ: other samples
    devs = [handle() for _ in range(IN_FLIGHT)]
    started, finished = {}, {}

    def job(k, i, ordered):
        started[i] = time.perf_counter()
        with ordered:  # (where a multi-rank caller issues the frame's reduce)
            pass
        out = frame(devs[k], i)
        finished[i] = time.perf_counter()
        return out

    pipe = D.FramePipeline(IN_FLIGHT)
    got = [f.result() for f in [pipe.submit(i, job) for i in range(FRAMES)]]
    pipe.close()
    for i in range(FRAMES):
        assert np.array_equal(got[i][0], want[i][0]), "frame %d: RGB8 differs" % i
        assert np.array_equal(bits(got[i][1]), bits(want[i][1])), "frame %d: accumulator differs" % i
    assert any(started[i + 1] < finished[i] for i in range(FRAMES - 1))  # and the frames did overlap
    # the same loop in C++ (host/frames.h through rth_render_frames: what a C++ host links), on the same handles
    px = api.render_frames(devs, FRAMES, TOTAL, 0, WORLD)
    for i in range(FRAMES):
        assert np.array_equal(px[i], want[i][0]), "C++ frame loop, frame %d" % i
    stats = [d.stats() for d in devs]
    assert sum(s["primary_samples"] for s in stats) == 2 * FRAMES * (TOTAL // WORLD) * W * H and all(s["dropped_paths"] == 0 for s in stats)
    for d in devs:
        d.close()
