"""z-slab sharding of the voxel grid across the GPUs of one node.

The reference already voxelizes independent 64^3 chunks from duplicated triangle lists with the voxel walk clamped
to the chunk (src/obj2voxel.cpp:226-243, src/voxelization.cpp:440-444); here the "chunk" is one GPU's z-slab.
Every output voxel is owned by exactly one slab, so ranks never exchange voxel data: the only collectives are the
sum of the per-slab voxel counts and the max of the per-rank times.
"""


def slab_range(rank, world, resolution):
    """Output-z range [z0, z1) owned by `rank`: equal heights, the last rank takes the remainder."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    h = resolution // world
    if h == 0:
        raise ValueError("more ranks than z layers")
    z0 = rank * h
    z1 = resolution if rank == world - 1 else z0 + h
    return z0, z1


def reduce_job(dist, count, seconds, device=None):
    """Whole-job voxel count (sum over ranks) and job time (max over ranks). `dist` is torch.distributed or None."""
    if dist is None or not dist.is_initialized():
        return int(count), float(seconds)
    import torch
    c = torch.tensor([float(count)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(c.item()), float(t.item())
