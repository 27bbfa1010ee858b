"""Diagnostic (not a test): two ranks time-sharing ONE GPU over gloo, the step replayed from GraphedTrainStep's two graphs --
where does a non-finite value first appear?
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 tests/diag_two_rank_graph.py 4
Found at the end of round 3: on some boxes the second replay of the forward + backward graph left ~95 % of ONE rank's weight
gradients non-finite although its loss was finite and every stage is followed by a device synchronisation (5 of 16 runs).  Cause:
nfl_composite_backward zeroed d_gmax with hipMemsetAsync; captured into a graph that is a memset NODE, and with a second process
replaying graphs on the same GPU it took effect out of order with the kernel after it (the atomicMax results wiped -> loss scale
from an all-zero maximum -> overflow).  Zeroed by a kernel instead: 0 of 16 runs.  Kept as the reproducer."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
import nerf_fl_amd
from nerf_fl_amd import parallel, synth
from nerf_fl_amd.train import Adam, GraphedTrainStep
models, emb = bench.build_models(dev, "cfg1")
torch.manual_seed(1234 + rank)
R = 4096
rays = synth.make_rays(R, 100 + rank).to(dev); ts = torch.zeros(R, dtype=torch.long, device=dev); target = torch.rand(R, 3, device=dev)
params = [p for m in models.values() for p in m.parameters()]
opt = Adam(params, lr=5e-4, eps=1e-8, capturable=True)
arena = parallel.GradArena(params)
g = GraphedTrainStep(models, emb, params, opt, None, rays, ts, target, 64, 64, white_back=True, all_reduce=True, arena=arena)
bad = lambda t: int((~torch.isfinite(t)).sum())
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    g.opt.sync_hyper()
    g.graph.replay(); torch.cuda.synchronize()
    b1 = bad(arena.flat); loss = float(g.out[0])
    arena.all_reduce(); torch.cuda.synchronize()
    b2 = bad(arena.flat)
    g.graph_opt.replay(); torch.cuda.synchronize()
    g.opt.note_replay(); torch.autograd.graph.increment_version(params)
    b3 = sum(bad(p) for p in params)
    print(f"rank {rank} it {it}: loss {loss:.5f} nonfinite grads after bwd {b1}, after all-reduce {b2}, params after adam {b3}", flush=True)
dist.barrier(); dist.destroy_process_group()
