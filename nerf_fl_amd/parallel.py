"""Ray-batch data parallelism: one process per GPU, RCCL over xGMI.

The reference's only parallelism is Lightning DDP over rays (train.py:235-236): each
process renders its own batch and DDP averages every parameter gradient with bucketed
all-reduces.  Rays are independent, so here each rank renders a contiguous shard with
no data-path collective, and the gradients (two fields + latent tables, ~1.4 M fp32,
5.5 MB) are averaged with ONE all-reduce over a single flat buffer: at this size the
collective is latency-bound, so one call beats DDP's several buckets.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced [lo, hi) of `n_items` for `rank` (first `n % world` ranks get one more)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rays(rays, ts=None, rank=None, world=None):
    """This rank's contiguous slice of a ray batch (and of its per-ray `ts`)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(rays.shape[0], rank, world)
    return (rays[lo:hi], None if ts is None else ts[lo:hi])


def all_reduce_gradients(params, group=None, average=True):
    """Average (or sum) the .grad of `params` across ranks with one flat all-reduce.
    Parameters whose grad is None contribute zeros (as DDP does for unused parameters)."""
    params = [p for p in params if p.requires_grad]
    if not params or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat.div_(world)
    views, off = [], 0
    for g in grads:
        views.append(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
    for p, g in zip(params, grads):
        if p.grad is None:
            p.grad = g                      # the zeros created above; filled by the batched copy below
    torch._foreach_copy_([p.grad for p in params], views)       # one multi-tensor kernel instead of ~50 copies
