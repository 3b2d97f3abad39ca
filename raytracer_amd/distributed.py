"""Sample-split data parallelism over the GPUs of one node.

The reference is single-device (one cl_context, template/precomp.h:1544-1546); frames shard naturally by SAMPLE:
every rank holds the whole scene and covers all pixels, rank r renders the global sample indices
first + r, first + r + world, ... and accumulates them into its own float4[W*H] buffer in HBM.  The random stream of
(pixel, sample) is a pure function of the pixel's seed and the global sample index, so the image does not
depend on the number of ranks.  The only exchange is ONE sum-reduce of the accumulators onto rank 0 at the end
(torch.distributed: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests), after which rank 0
finalises with f = total samples (RGBF32_to_RGB8, precomp.h:449-465).

One process per GPU; no data-path collective inside the bounce loop.
"""
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
