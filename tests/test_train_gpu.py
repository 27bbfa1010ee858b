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
