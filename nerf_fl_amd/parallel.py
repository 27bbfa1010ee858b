"""Ray-batch data parallelism: one process per GPU, RCCL over xGMI.

The reference's only parallelism is Lightning DDP over rays (train.py:235-236): each
process renders its own batch and DDP averages every parameter gradient with bucketed
all-reduces.  Rays are independent, so here each rank renders a contiguous shard with
no data-path collective, and the gradients (two fields + latent tables, ~1.4 M fp32,
5.5 MB) are averaged with ONE all-reduce: at this size the collective is latency-bound,
so one call beats DDP's several buckets.

`GradArena` is the memory that all-reduce runs on: one flat fp32 buffer that holds the
gradient of every trainable parameter, `p.grad` being views into it.  The HIP backward
(`render_rays(..., grad_arena=arena)`) writes the weight gradients of both fields and the
latent-table gradients straight into it -- no per-step gradient allocation, no `cat`
before and no copy back after the collective, and a fixed address that a captured HIP
graph can replay on.
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


class GradArena:
    """One contiguous fp32 buffer for the gradients of `params` (every tensor 16-byte aligned inside it); `p.grad` of
    each parameter is a view into it.  Pass it to `render_rays(..., grad_arena=arena)`: the backward then WRITES the
    gradients of the parameters it owns into their views (overwriting: one render_rays call per optimizer step) instead
    of returning fresh tensors for autograd to accumulate, and `all_reduce()` averages all of them in place with one
    collective.  Parameters the backward does not reach (e.g. a transient head an `output_transient=False` call leaves
    out) keep the zeros `zero()` left."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradArena: no trainable parameters")
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("GradArena: fp32 parameters on one device only")
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._views = {id(p): self.flat[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, offs)}
        self.attach()

    def view(self, p):
        """The arena's gradient view of parameter `p`, or None if `p` is not in the arena."""
        return self._views.get(id(p))

    def attach(self):
        """(Re-)point every p.grad at its view (e.g. after zero_grad(set_to_none=True) dropped them)."""
        for p in self.params:
            v = self._views[id(p)]
            if p.grad is not v:
                p.grad = v

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, group=None, average=True, force=False):
        """Average (or sum) the whole arena across ranks: ONE in-place collective on memory the backward already owns.
        `force`: issue the collective at world size 1 as well (exercises the RCCL path on a single GPU)."""
        if not dist.is_initialized():
            return
        world = dist.get_world_size(group)
        if world == 1 and not force:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if average and world > 1:
            self.flat.div_(world)


def all_reduce_gradients(params, group=None, average=True, arena=None, force=False):
    """Average (or sum) the .grad of `params` across ranks with one flat all-reduce.  With a `GradArena` that holds
    them this is a single in-place collective; without one the gradients are gathered into a temporary flat buffer and
    copied back (parameters whose grad is None contribute zeros, as DDP does for unused parameters)."""
    if arena is not None:
        arena.all_reduce(group=group, average=average, force=force)
        return
    params = [p for p in params if p.requires_grad]
    if not params or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    world = dist.get_world_size(group)
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average and world > 1:
        flat.div_(world)
    views, off = [], 0
    for g in grads:
        views.append(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
    for p, g in zip(params, grads):
        if p.grad is None:
            p.grad = g                      # the zeros created above; filled by the batched copy below
    torch._foreach_copy_([p.grad for p in params], views)       # one multi-tensor kernel instead of ~50 copies
