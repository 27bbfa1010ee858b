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
