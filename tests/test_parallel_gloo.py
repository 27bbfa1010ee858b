"""world_size-2 CPU (gloo) test of the ray sharding and the flat gradient all-reduce."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_fl_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    rays = torch.arange(101 * 8, dtype=torch.float32).reshape(101, 8)
    ts = torch.arange(101)
    r, t = parallel.shard_rays(rays, ts)
    # a toy "field": gradient of sum(w * rays) w.r.t. w is the column sum of the local shard
    w = torch.nn.Parameter(torch.ones(8))
    b = torch.nn.Parameter(torch.zeros(3))           # unused parameter: grad None on every rank
    (r * w).sum().backward()
    parallel.all_reduce_gradients([w, b])
    out[rank] = (r.shape[0], int(t[0]), int(t[-1]), w.grad.clone(), b.grad.clone())
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 101, 4096):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flat_allreduce_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    rays = torch.arange(101 * 8, dtype=torch.float32).reshape(101, 8)
    assert out[0][0] + out[1][0] == 101 and out[0][0] == 51
    assert out[0][1] == 0 and out[0][2] == 50 and out[1][1] == 51 and out[1][2] == 100
    expect = rays.sum(0) / world                     # average of the two shards' column sums
    for r in range(world):
        assert torch.allclose(out[r][3], expect)
        assert torch.equal(out[r][4], torch.zeros(3))


def _arena_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # three parameters of awkward sizes (the arena aligns each to 16 bytes); one of them is never reached by "backward"
    ps = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2, 2))]
    arena = parallel.GradArena(ps)
    assert all(p.grad is arena.view(p) for p in ps)
    assert all(p.grad.data_ptr() % 16 == 0 for p in ps)
    ptr0 = arena.flat.data_ptr()
    for step in range(3):                    # the "backward" writes in place, the collective runs in place
        arena.view(ps[0]).fill_(float(rank + 1 + step))
        arena.view(ps[1]).copy_(torch.arange(7.0) * (rank + 1))
        for p in ps:
            p.grad = None                    # an optimizer's zero_grad(set_to_none=True) in between ...
        arena.attach()                       # ... is undone by attach()
        parallel.all_reduce_gradients(ps, arena=arena)
        assert arena.flat.data_ptr() == ptr0 and all(p.grad is arena.view(p) for p in ps)
    out[rank] = [p.grad.clone() for p in ps]
    dist.destroy_process_group()


def test_grad_arena_allreduce_world2():
    """GradArena: p.grad are views of ONE flat buffer; the all-reduce averages it in place (no cat, no copy back)."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_arena_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        assert torch.equal(out[r][0], torch.full((5, 3), (3 + 4) / 2.0))          # step 2: ranks wrote 3 and 4
        assert torch.allclose(out[r][1], torch.arange(7.0) * 1.5)
        assert torch.equal(out[r][2], torch.zeros(2, 2))                           # never written: stays zero


def test_grad_arena_rejects_mixed_inputs():
    import pytest
    with pytest.raises(ValueError):
        parallel.GradArena([torch.nn.Parameter(torch.zeros(3), requires_grad=False)])
    with pytest.raises(ValueError):
        parallel.GradArena([torch.nn.Parameter(torch.zeros(3, dtype=torch.float64))])
