"""Multi-GPU frame rendering: one process per GPU, tiles dealt round-robin.

The scene is replicated on every rank.  Rank r of N renders the 32x32 tiles t
with t % N == r into a zeroed full-size framebuffer (``render_shard``), and one
``reduce(sum)`` to rank 0 assembles the frame -- the only collective of the
path.  Tiles are disjoint, so the sum adds each pixel to zeros and the result
is bit-identical to a single-GPU frame.
"""
import os


def env_rank():
    """(rank, world_size, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def tiles_of(width, height, rank, world):
    """Indices of the 32x32 tiles (row-major) that ``rank`` owns."""
    tiles_x, tiles_y = (width + 31) // 32, (height + 31) // 32
    return list(range(rank, tiles_x * tiles_y, world))


def render_frame(render_shard, framebuffer, rank, world, dist=None, resolve=None, single_rank_collective=False):
    """``render_shard(shard_index, shard_count, framebuffer)`` fills this rank's
    tiles (and zeroes the rest); with world > 1 the buffers are summed onto
    rank 0 through ``dist`` (torch.distributed, backend nccl = RCCL on GPUs,
    gloo in CPU tests).  Returns what ``render_shard`` returned.

    With a reconstruction filter a sample also weighs on pixels of other ranks'
    tiles: ``render_shard`` must then leave a raw film (RT_RENDER_RAW_FILM:
    sums of w*L and of w, which add across ranks) and ``resolve(framebuffer)``
    normalises the summed film on rank 0 (rt_film_resolve_dev).

    ``single_rank_collective`` runs the reduce through ``dist`` for a world of one as well (a
    rehearsal of the collective's code path where only one device is at hand)."""
    out = render_shard(rank, world, framebuffer)
    if world > 1 or single_rank_collective:
        dist.reduce(framebuffer, dst=0, op=dist.ReduceOp.SUM)
    if resolve is not None and rank == 0:
        resolve(framebuffer)
    return out
