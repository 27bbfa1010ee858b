"""The training harness (role of train.py's NeRFSystem): a few epochs on a tiny seeded scene raise the
validation PSNR, and checkpoints round-trip through the reference's key prefixes."""
import pytest
import torch

from oracle import nerfw_oracle as orc

pytestmark = pytest.mark.gpu


def test_fit_and_checkpoint(tmp_path):
    import gpu_util
    from nerf_fl_amd.train import RayTrainer
    dev = gpu_util.DEV
    spec = orc.FieldSpec("coarse")
    teacher = orc.make_field_params(spec, 21, "sharp")
    rays, val = orc.make_rays(4096, 31), orc.make_rays(512, 32)
    with torch.no_grad():
        kw = dict(n_samples=48, white_back=True, noise_std=0.0)
        rgb = orc.render_rays(spec, teacher, None, None, rays, **kw)["rgb_coarse"]
        vrgb = orc.render_rays(spec, teacher, None, None, val, **kw)["rgb_coarse"]
    tr = RayTrainer(dev, N_samples=32, N_importance=32, encode_a=True, encode_t=True, N_vocab=8, batch_size=512,
                    lr_scheduler="cosine", num_epochs=3)
    ts = torch.randint(0, 8, (4096,), device=dev)
    vts = torch.zeros(512, dtype=torch.long, device=dev)
    rays, rgb, val, vrgb = rays.to(dev), rgb.to(dev), val.to(dev), vrgb.to(dev)
    p0 = tr.validate(val, vrgb, vts)
    for _ in range(3):
        loss, train_psnr = tr.fit_epoch(rays, rgb, ts)
        assert loss == loss                      # not NaN
    p1 = tr.validate(val, vrgb, vts)
    print("PSNR before / after 3 epochs:", p0, p1)
    assert p1 > p0 + 3.0, (p0, p1)

    # checkpoint keys carry the reference's prefixes (train.py:51-76; utils/__init__.py:67-88)
    sd = tr.state_dict()
    assert "nerf_coarse.xyz_encoding_1.0.weight" in sd and "nerf_fine.transient_beta.0.bias" in sd
    assert "embedding_a.weight" in sd and "embedding_t.weight" in sd
    path = str(tmp_path / "ckpt" / "epoch=2.ckpt")
    tr.save(path, epoch=2)
    tr2 = RayTrainer(dev, N_samples=32, N_importance=32, encode_a=True, encode_t=True, N_vocab=8, batch_size=512, seed=5)
    tr2.load(path)
    assert abs(tr2.validate(val, vrgb, vts) - p1) < 1e-4



def test_trainer_with_graphed_steps_fits_like_the_eager_one():
    """RayTrainer(use_graph=True): every step of fit_epoch replayed from one captured HIP graph (what the README batch of 1024
    rays wants: the launches of an eager step are its critical path there) trains like the eager trainer."""
    import gpu_util
    from nerf_fl_amd.train import RayTrainer
    dev = gpu_util.DEV
    spec = orc.FieldSpec("coarse")
    teacher = orc.make_field_params(spec, 21, "sharp")
    rays, val = orc.make_rays(4096, 31), orc.make_rays(512, 32)
    with torch.no_grad():
        kw = dict(n_samples=48, white_back=True, noise_std=0.0)
        rgb = orc.render_rays(spec, teacher, None, None, rays, **kw)["rgb_coarse"]
        vrgb = orc.render_rays(spec, teacher, None, None, val, **kw)["rgb_coarse"]
    ts = torch.randint(0, 8, (4096,), device=dev)
    vts = torch.zeros(512, dtype=torch.long, device=dev)
    rays, rgb, val, vrgb = rays.to(dev), rgb.to(dev), val.to(dev), vrgb.to(dev)
    out = {}
    for graph in (False, True):
        tr = RayTrainer(dev, N_samples=32, N_importance=32, encode_a=True, encode_t=True, N_vocab=8, batch_size=512,
                        lr_scheduler="cosine", num_epochs=3, use_graph=graph)
        p0 = tr.validate(val, vrgb, vts)
        for _ in range(3):
            loss, _psnr = tr.fit_epoch(rays, rgb, ts)
            assert loss == loss
        out[graph] = (p0, tr.validate(val, vrgb, vts))
    print("PSNR before / after 3 epochs, eager:", out[False], "graphed:", out[True])
    assert out[True][1] > out[True][0] + 3.0
    # 24 steps from scratch with different random draws (Philox offsets; the capture's warm-up takes two extra steps on the
    # first batch): both must have learnt, and the graphed trainer must not trail the eager one
    assert out[False][1] > out[False][0] + 3.0 and out[True][1] > out[False][1] - 3.0


def test_one_launch_adam_matches_torch_adam():
    """nerf_fl_amd.train.Adam (C ABI nfl_adam_step) against torch.optim.Adam on identical parameters and gradients,
    over several steps, with a learning-rate change and a parameter that gets no gradient in one step."""
    import gpu_util
    from nerf_fl_amd.train import Adam
    dev = gpu_util.DEV
    g = torch.Generator().manual_seed(11)
    shapes = [(256, 63), (256,), (3, 128), (1,), (100, 48)] + [(17, 5)] * 70       # > 64 tensors: two launches
    ref = [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref, o_mine = torch.optim.Adam(ref, lr=5e-4, eps=1e-8), Adam(mine, lr=5e-4, eps=1e-8)
    for step in range(6):
        if step == 3:
            for o in (o_ref, o_mine):
                o.param_groups[0]["lr"] = 1e-4
        for k, (a, b) in enumerate(zip(ref, mine)):
            if step == 2 and k == 1:
                a.grad = b.grad = None
                continue
            gr = (torch.randn(*a.shape, generator=g) * 10.0 ** float(torch.randint(-6, 2, (1,), generator=g))).to(dev)
            a.grad, b.grad = gr.clone(), gr.clone()
        o_ref.step()
        o_mine.step()
    for a, b in zip(ref, mine):
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item())
    sd = o_mine.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    torch.optim.Adam(mine, lr=5e-4).load_state_dict(sd)          # interchangeable state


def test_one_launch_adam_bumps_parameter_versions():
    """render_rays re-packs its weight streams when a parameter's version counter moves; an optimiser that writes the
    parameters from a kernel has to move it."""
    import gpu_util
    from nerf_fl_amd.train import Adam
    p = torch.nn.Parameter(torch.ones(8, device=gpu_util.DEV))
    o = Adam([p], lr=0.1)
    p.grad = torch.ones_like(p)
    v0 = p._version
    o.step()
    assert p._version > v0 and not torch.equal(p.detach().cpu(), torch.ones(8))


@pytest.mark.parametrize("variant", ["coarse_only", "base", "nerfw"])
def test_fused_nerfw_loss_matches_reference_formula(variant):
    """train.NerfWLoss on device tensors (C ABI nfl_loss_forward / nfl_loss_backward) against the oracle's restatement
    of losses.py:35-50 with autograd, values and gradients, with unequal weights on the terms."""
    import gpu_util
    from nerf_fl_amd.train import NerfWLoss
    dev = gpu_util.DEV
    g = torch.Generator().manual_seed(5)
    R, F = 333, 40
    inp = {"rgb_coarse": torch.rand(R, 3, generator=g)}
    if variant != "coarse_only":
        inp["rgb_fine"] = torch.rand(R, 3, generator=g)
    if variant == "nerfw":
        inp["beta"] = torch.rand(R, generator=g) * 0.5 + 0.1
        inp["transient_sigmas"] = torch.rand(R, F, generator=g) * 3
    target = torch.rand(R, 3, generator=g)
    wts = {"c_l": 1.0, "f_l": 0.7, "b_l": 1.3, "s_l": 2.0}
    ref_in = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    ref = orc.nerfw_loss(ref_in, target)
    sum(wts[k] * v for k, v in ref.items()).backward()
    hip_in = {k: v.to(dev).requires_grad_(True) for k, v in inp.items()}
    out = NerfWLoss()(hip_in, target.to(dev))
    assert list(out.keys()) == list(ref.keys())
    sum(wts[k] * v for k, v in out.items()).backward()
    for k in ref:
        assert abs(out[k].item() - ref[k].item()) <= 2e-6 * max(1.0, abs(ref[k].item())), k
    for k in inp:
        e, r = (hip_in[k].grad.cpu() - ref_in[k].grad).abs().max().item(), ref_in[k].grad.abs().max().item()
        assert e <= 2e-6 * r + 1e-12, (k, e, r)


def test_capturable_adam_equals_the_eager_one():
    """Adam(capturable=True) (C ABI nfl_adam_step_dev: lr / betas / eps / step count read from device memory) against
    the by-value launch over 5 steps with a learning-rate change in between."""
    import gpu_util
    from nerf_fl_amd.train import Adam
    dev = gpu_util.DEV
    g = torch.Generator().manual_seed(3)
    shapes = [(256, 63), (256,), (128, 283), (3,)]
    base = [torch.randn(*s, generator=g) for s in shapes]
    a = [torch.nn.Parameter(b.clone().to(dev)) for b in base]
    b = [torch.nn.Parameter(b.clone().to(dev)) for b in base]
    oa, ob = Adam(a, lr=5e-4, eps=1e-8), Adam(b, lr=5e-4, eps=1e-8, capturable=True)
    for step in range(5):
        if step == 3:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 2e-4
        for x, y in zip(a, b):
            gr = torch.randn(*x.shape, generator=g).to(dev)
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    for x, y in zip(a, b):
        assert (x - y).abs().max().item() <= 1e-6 * max(1.0, x.abs().max().item())
    assert ob.state[b[0]]["step"] == 5


def test_graphed_train_step_trains_like_the_eager_step():
    """GraphedTrainStep (re-pack + both passes + fused loss + backward + Adam captured in one HIP graph): replays draw
    fresh random numbers, advance Adam's device-side step count, follow learning-rate changes, and fit the scene as the
    eager loop does (same data, same number of steps; the draws differ, so PSNRs are compared, not weights)."""
    import gpu_util
    from nerf_fl_amd import NeRF, PosEmbedding, render_rays
    from nerf_fl_amd.train import Adam, GraphedTrainStep, psnr
    dev = gpu_util.DEV
    spec = orc.FieldSpec("coarse")
    teacher = orc.make_field_params(spec, 21, "sharp")
    R, S, I, STEPS = 512, 32, 32, 60
    rays = orc.make_rays(4096, 31)
    with torch.no_grad():
        rgb = orc.render_rays(spec, teacher, None, None, rays, n_samples=32, white_back=True, noise_std=0.0)["rgb_coarse"]
    rays, rgb = rays.to(dev), rgb.to(dev)
    ts = torch.zeros(R, dtype=torch.long, device=dev)
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}

    def fresh():
        models = {"coarse": gpu_util.module_from(spec, orc.make_field_params(spec, 22, "default")),
                  "fine": gpu_util.module_from(orc.FieldSpec("fine"), orc.make_field_params(orc.FieldSpec("fine"), 23, "default"))}
        return models, [p for m in models.values() for p in m.parameters()]

    def val(models):
        with torch.no_grad():
            out = render_rays(models, emb, rays[:1024], torch.zeros(1024, dtype=torch.long, device=dev), S, False, 0, 0.0, I,
                              32768, True, False)
        return float(psnr(out["rgb_fine"], rgb[:1024]))

    gen = torch.Generator().manual_seed(4)
    batches = [torch.randint(0, 4096, (R,), generator=gen).to(dev) for _ in range(STEPS + 2)]
    # eager
    models, params = fresh()
    p0 = val(models)
    opt = Adam(params, lr=1e-3)
    for it in range(STEPS + 2):
        opt.zero_grad(set_to_none=True)
        idx = batches[it]
        render_rays(models, emb, rays[idx], ts, S, False, 1.0, 1.0, I, 32768, True, False, loss_target=rgb[idx])["_nerfw_loss"].backward()
        opt.step()
    p_eager = val(models)
    # graphed: 2 warm-up steps inside the constructor on batch 0, then STEPS replays
    models, params = fresh()
    opt = Adam(params, lr=1e-3, capturable=True)
    g = GraphedTrainStep(models, emb, params, opt, None, rays[batches[0]], ts, rgb[batches[0]], S, I, warmup=2)
    losses = []
    for it in range(STEPS):
        idx = batches[it + 2]
        g.load(rays[idx], ts, rgb[idx])
        if it == STEPS // 2:
            opt.param_groups[0]["lr"] = 5e-4              # picked up by the captured launch through sync_hyper()
        loss, _ = g.replay()
        losses.append(loss.clone())
    p_graph = val(models)                                   # eager render after replays: the weight streams must re-pack
    losses = torch.stack(losses).cpu()
    print(f"PSNR before {p0:.2f} dB, eager-trained {p_eager:.2f} dB, graph-trained {p_graph:.2f} dB")
    assert torch.isfinite(losses).all() and len(set(losses.tolist())) > STEPS // 2, "replays must see new batches and draws"
    assert opt.state[params[0]]["step"] == STEPS + 2
    assert p_eager > p0 + 3.0 and p_graph > p0 + 3.0
    assert abs(p_eager - p_graph) <= 1.5
