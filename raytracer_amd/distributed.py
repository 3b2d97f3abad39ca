"""Sample-split data parallelism over the GPUs of one node.

The reference is single-device (one cl_context, template/precomp.h:1544-1546); frames shard naturally by SAMPLE:
every rank holds the whole scene and covers all pixels, rank r renders the global sample indices
first + r, first + r + world, ... and accumulates them into its own float4[W*H] buffer in HBM.  The random stream of
(pixel, sample) is a pure function of the pixel's seed and the global sample index, so the image does not
depend on the number of ranks.  The only exchange is ONE sum-reduce of the accumulators onto rank 0 at the end
(torch.distributed: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests), after which rank 0
finalises with f = total samples (RGBF32_to_RGB8, precomp.h:449-465).

One process per GPU; no data-path collective inside the bounce loop.

Strong scaling (one frame split over the ranks, delivered, then the next: BASELINE.json configs[3]) adds FramePipeline: the end
of a frame -- a handful of paths with thousands of bounces left, a serial chain that no amount of GPU shortens -- runs beside the
bulk of the following frames instead of holding the GPU alone, by giving consecutive frames to different handles.
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def split_samples(first, count, rank, world):
    """(sample_begin, sample_count, sample_stride) for rtmi_render on `rank`: the indices first + rank + k * world
    below first + count.  The union over ranks is exactly [first, first + count), each index once."""
    if world < 1 or not 0 <= rank < world or count < 0 or first < 0:
        raise ValueError("bad split")
    mine = (count - rank + world - 1) // world if count > rank else 0
    return first + rank, mine, world


def render_split(render, first, count, rank=None, world=None):
    """Calls render(sample_begin, sample_count, sample_stride) for this rank's share; returns the share size."""
    if rank is None or world is None:
        rank, world = world_info()
    begin, mine, stride = split_samples(first, count, rank, world)
    if mine > 0:
        render(begin, mine, stride)
    return mine


def reduce_accumulator(acc, dst=0):
    """Sum the per-rank accumulators onto `dst` (one collective per render).  `acc` is a torch tensor that aliases
    the buffer the kernels accumulate into (rtmi_set_accumulator).  No-op for a single process."""
    _, world = world_info()
    if world > 1:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM)
    return acc


class FramePipeline:
    """Frames in flight on `n` handles of one GPU.  A frame is rendered, finished and delivered entirely by ONE handle (its own
    accumulator, late sums, queues and streams: per frame exactly the calls an unpipelined caller makes, so every delivered
    frame has the unpipelined frame's bits); consecutive frames go to consecutive handles, each driven by a host thread of its
    own, so that while handle k finishes the last paths of frame i (rtmi_flush .. rtmi_finalize: few waves, long latency) handle
    k+1 is in the bulk of frame i+1.  The collectives of the frames (one reduce each) are issued in frame order on every rank:
    `job( k, i, ordered )` wraps its reduce in `with ordered:`.

        pipe = FramePipeline(3)
        futures = [pipe.submit(i, job) for i in range(frames)]     # job(k, i, ordered) -> the frame; k = i % n, the handle to use
        frames = [f.result() for f in futures]                     # in order
    """

    class _Turn:
        def __init__(self, pipe, i):
            self.pipe, self.i = pipe, i

        def __enter__(self):
            with self.pipe._cv:
                self.pipe._cv.wait_for(lambda: self.pipe._turn == self.i)

        def __exit__(self, *exc):
            with self.pipe._cv:
                self.pipe._turn = self.i + 1
                self.pipe._cv.notify_all()
            return False

    def __init__(self, n):
        if n < 1:
            raise ValueError("at least one handle")
        self.n = n
        self._workers = [ThreadPoolExecutor(1) for _ in range(n)]  # one thread per handle: a handle's frames run one after the other
        self._cv = threading.Condition()
        self._turn = 0
        self._next = 0

    def submit(self, i, job):
        """frame i (submitted in order 0, 1, 2, ...) on handle i % n; returns a future of job's result"""
        if i != self._next:
            raise ValueError("frames are submitted in order")
        self._next += 1

        def run():
            turn = FramePipeline._Turn(self, i)
            took = [False]

            class Ordered:
                def __enter__(s):
                    turn.__enter__()
                    took[0] = True

                def __exit__(s, *exc):
                    return turn.__exit__(*exc)
            try:
                return job(i % self.n, i, Ordered())
            finally:
                if not took[0]:  # a job without a collective (one rank), or one that failed before it: the turn still passes on
                    turn.__enter__()
                    turn.__exit__()
        return self._workers[i % self.n].submit(run)

    def close(self):
        for w in self._workers:
            w.shutdown(wait=True)
