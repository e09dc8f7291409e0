"""Data-parallel plumbing: frames (pairs) shard across ranks, shared mesh-parameter gradients are all-reduced.

One process per GPU; backend 'nccl' is RCCL on ROCm (xGMI inside a node), 'gloo' in the CPU tests.  The reference
gets this from DistributedSampler + DistributedDataParallel (/root/reference/dataloader/vid.py:126-131,
nnutils/train_utils.py:104-109); the full model goes through DDP here too (nnutils/train_utils.py of this repo),
while bench.py's rasteriser workload uses the two helpers below directly.
"""
import torch
import torch.distributed as dist


def shard(items, rank, world):
    """Round-robin split used by DistributedSampler: rank r takes items r, r+world, ... (after padding the list by
    wrapping around so that every rank gets the same count)."""
    items = list(items)
    if not items:
        return []
    per = -(-len(items) // world)
    padded = (items * (per * world // len(items) + 1))[:per * world]
    return padded[rank::world]


def allreduce_grads_(tensors, average=True, group=None):
    """In-place all-reduce of a list of gradient tensors as ONE flat message (the mesh gradients are a few tens of
    KB: latency-bound, so a single collective instead of one per tensor).  Returns the tensors."""
    tensors = [t for t in tensors if t is not None]
    if not tensors or not (dist.is_available() and dist.is_initialized()):
        return tensors
    world = dist.get_world_size(group)
    if world == 1:
        return tensors
    if len(tensors) == 1 and tensors[0].is_contiguous():                     # already one flat message: reduce it in place
        dist.all_reduce(tensors[0], group=group)
        if average:
            tensors[0] /= world
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, group=group)
    if average:
        flat /= world
    o = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[o:o + n].view_as(t))
        o += n
    return tensors


def allreduce_grads_async(tensors, average=True, group=None):
    """allreduce_grads_ started on the process group's communication stream; returns a function that waits for it and
    scatters the reduced values back into the tensors.  The caller runs other kernels in between (the rest of the backward
    pass): with RCCL the collective's kernels then overlap them, as DDP's bucket hooks do in the reference."""
    tensors = [t for t in tensors if t is not None]
    if not tensors or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return lambda: tensors
    world = dist.get_world_size(group)
    flat = torch.cat([t.reshape(-1) for t in tensors])
    work = dist.all_reduce(flat, group=group, async_op=True)

    def finish():
        work.wait()
        if average:
            flat.div_(world)
        o = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[o:o + n].view_as(t))
            o += n
        return tensors
    return finish
