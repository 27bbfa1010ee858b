"""The gradient arena and the RCCL path on ONE GPU (reference train.py:235-236 runs DDP over NCCL).

  * `render_rays(..., grad_arena=arena)`: the HIP backward writes the gradients of both fields and of the latent tables
    straight into the arena's views -- the same numbers the allocate-and-return path produces;
  * `bench.py --gpus 1 --force-dist`: process group with backend nccl (= RCCL) at world size 1, the in-place all-reduce
    on the arena issued for real, eager and captured inside the step's HIP graph;
  * Adam(capturable=True) after load_state_dict; GraphedTrainStep with NerfWLoss constants other than the defaults.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(name):
    import gpu_util
    cfg, a = gu.load(name)
    (spec_c, P_c, spec_f, P_f), kw = gu.oracle_kwargs(cfg, a)
    dev = gpu_util.DEV
    models = {"coarse": gpu_util.module_from(spec_c, P_c), "fine": gpu_util.module_from(spec_f, P_f)}
    emb = gpu_util.make_embeddings(spec_c.n_emb_xyz, False)
    for k, dim in (("a", cfg.get("n_a", 48)), ("t", cfg.get("n_tau", 16))):
        if kw.get(k + "_emb") is not None:
            table = gu.embedding_table(cfg, k)
            e = torch.nn.Embedding(table.shape[0], dim).to(dev)
            e.weight.data.copy_(table)
            emb[k] = e
    extra = {k: kw[k].to(dev) for k in ("perturb_rand", "noise_coarse", "u", "noise_fine") if kw.get(k) is not None}
    rays, ts, target = a["rays"].to(dev), a["ts"].to(dev), a["target"].to(dev)
    params = [p for m in list(models.values()) + [emb[k] for k in ("a", "t") if k in emb] for p in m.parameters()]
    args = (models, emb, rays, ts, cfg["S"], cfg["use_disp"], cfg["perturb"], cfg["noise_std"], cfg["I"], 32768,
            cfg["white_back"], False)
    return args, extra, target, params


@pytest.mark.parametrize("name", ["g11_grad_cfg2", "g15_photo_stoch"])
def test_arena_gradients_equal_returned_gradients(name):
    from nerf_fl_amd import parallel, render_rays
    args, extra, target, params = _setup(name)
    for p in params:
        p.grad = None
    render_rays(*args, loss_target=target, **extra)["_nerfw_loss"].backward()
    ref = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    arena = parallel.GradArena(params)
    ptr = arena.flat.data_ptr()
    for _ in range(2):                       # twice: the second call must overwrite, not accumulate
        for p in params:
            p.grad = None                    # the backward re-attaches the views
        render_rays(*args, loss_target=target, grad_arena=arena, **extra)["_nerfw_loss"].backward()
    assert arena.flat.data_ptr() == ptr
    scale = max(r.abs().max().item() for r in ref)
    for p, r in zip(params, ref):
        assert p.grad is arena.view(p)
        # same kernels on the same inputs: what differs is the summation order of the fp32 atomics
        assert (p.grad - r).abs().max().item() <= 2e-4 * max(r.abs().max().item(), 1e-3 * scale)


def _bench(extra_args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--no-extras", "--sustained-seconds", "0", "--force-dist"] + extra_args,
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_rccl_world1_eager_step():
    """backend nccl IS RCCL on ROCm: init_process_group(device_id=...), one in-place all-reduce of the 1.2 M-float
    gradient arena per step, Adam, re-pack -- on the one GPU of this box."""
    out = _bench([])
    assert out["backend"] == "nccl" and out["n_gpus"] == 1
    assert out["all_reduce"]["forced_at_world_1"] is True and out["all_reduce"]["numel"] >= 2 * 595844
    assert out["replica_param_max_diff"] == 0.0 and out["value"] > 0


def test_rccl_world1_graphed_step():
    """The same step replayed from HIP graphs: the collective is recorded inside the one graph."""
    out = _bench(["--graph"])
    assert out["backend"] == "nccl" and out["config"]["hip_graph"] is True
    assert out["all_reduce"]["captured_in_graph"] is True
    assert out["value"] > 0


def test_capturable_adam_follows_a_loaded_checkpoint():
    """Adam(capturable=True): load_state_dict into an optimizer that has already stepped -- the device-side step counter
    must restart from the checkpoint's count, i.e. continue exactly like the by-value optimizer does."""
    import gpu_util
    from nerf_fl_amd.train import Adam
    dev = gpu_util.DEV
    g = torch.Generator().manual_seed(9)
    base = [torch.randn(64, 33, generator=g), torch.randn(64, generator=g)]
    grads = [[torch.randn(*b.shape, generator=g).to(dev) for b in base] for _ in range(7)]

    def run(opt, ps, steps):
        for s in steps:
            for p, gr in zip(ps, grads[s]):
                p.grad = gr.clone()
            opt.step()

    a = [torch.nn.Parameter(b.clone().to(dev)) for b in base]
    oa = Adam(a, lr=1e-2)
    run(oa, a, range(2))
    import copy
    ckpt_params, ckpt_opt = [p.detach().clone() for p in a], copy.deepcopy(oa.state_dict())    # state_dict() shares the tensors
    run(oa, a, range(2, 5))                                        # the continuation to reproduce
    b = [torch.nn.Parameter(x.clone().to(dev)) for x in base]
    ob = Adam(b, lr=1e-2, capturable=True)
    run(ob, b, range(5, 7))                                        # the capturable optimizer has its own history first
    with torch.no_grad():
        for p, c in zip(b, ckpt_params):
            p.copy_(c)
    ob.load_state_dict(ckpt_opt)
    run(ob, b, range(2, 5))
    for x, y in zip(a, b):
        assert (x - y).abs().max().item() <= 1e-6 * max(1.0, x.abs().max().item())
    assert ob.state[b[0]]["step"] == 5


def test_graphed_step_uses_the_given_loss_constants():
    """GraphedTrainStep(loss_coef, lambda_u) reach the fused loss (the graph used to capture the defaults)."""
    from nerf_fl_amd import render_rays
    from nerf_fl_amd.train import Adam, GraphedTrainStep
    args, extra, target, params = _setup("g15_photo_stoch")
    models, emb, rays, ts, S, use_disp, perturb, noise_std, I, _chunk, white_back, _tt = args
    with torch.no_grad():
        snap = [p.detach().clone() for p in params]
    opt = Adam(params, lr=0.0, capturable=True)                    # lr 0: the weights stay, only the loss is looked at
    g = GraphedTrainStep(models, emb, params, opt, None, rays, ts, target, S, I, use_disp=use_disp, perturb=0.0,
                         noise_std=0.0, white_back=white_back, warmup=1, loss_coef=0.5, lambda_u=0.2)
    loss, _ = g.replay()
    for p, s in zip(params, snap):
        assert torch.equal(p.detach(), s)
    res = render_rays(models, emb, rays, ts, S, use_disp, 0.0, 0.0, I, 32768, white_back, False, loss_target=target,
                      loss_coef=0.5, lambda_u=0.2)
    dflt = render_rays(models, emb, rays, ts, S, use_disp, 0.0, 0.0, I, 32768, white_back, False, loss_target=target)
    want, other = res["_nerfw_loss"].item(), dflt["_nerfw_loss"].item()
    assert abs(want - other) > 1e-3 * abs(other)                   # the constants matter on this fixture
    assert abs(float(loss) - want) <= 1e-5 * max(1.0, abs(want))
