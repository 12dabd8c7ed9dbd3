"""
Frame sharding across the GPUs of one node (one process per GPU) and the final gather.

Replaces the ssh/pipe job farm of the reference (distribute.py:131-248): frames of an
animation are independent, so rank r renders frames r, r+world, r+2*world, ... with no
exchange in the data path; finished 8/16-bit frames are gathered to rank 0 with one
collective per round of frames (RCCL over xGMI when the backend is "nccl"; the same code
runs on "gloo" for the CPU tests).
"""
import torch
import torch.distributed as dist


def shard(items, rank=None, world=None):
    """Round-robin share of ``items`` for this rank (all of them when not distributed)."""
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
    return list(items)[rank::world]


def gather_frame(frame, dst=0):
    """
    Gather one same-shaped frame tensor from every rank to ``dst``.
    Returns the list of per-rank tensors on ``dst`` and None elsewhere.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [frame]
    rank, world = dist.get_rank(), dist.get_world_size()
    out = [torch.empty_like(frame) for _ in range(world)] if rank == dst else None
    dist.gather(frame, out, dst=dst)
    return out


def gather_animation(local_frames, nframes, dst=0):
    """
    ``local_frames``: this rank's frames in shard order (tensors of one shape).  Returns the
    full animation in frame order on ``dst`` (list of tensors), None elsewhere.  Ranks whose
    shard is one frame short contribute a dummy in the last round.
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    rounds = (nframes + world - 1) // world
    result = [None] * nframes if rank == dst else None
    proto = local_frames[0] if local_frames else None
    for k in range(rounds):
        f = local_frames[k] if k < len(local_frames) else torch.zeros_like(proto)
        got = gather_frame(f, dst)
        if rank == dst:
            for r, t in enumerate(got):
                idx = k * world + r
                if idx < nframes:
                    result[idx] = t
    return result
