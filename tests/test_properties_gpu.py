"""Full-size (BASELINE.json configs) checks through size-independent properties of the
renderer: the oracle is too slow at these sizes, but the domain offers invariants."""
import pytest
import torch

from oracle import nerfw_oracle as orc

pytestmark = pytest.mark.gpu


def _setup(fine, n_vocab=100, regime="sharp"):
    import gpu_util
    from nerf_fl_amd import PosEmbedding
    spec_c = orc.FieldSpec("coarse")
    models = {"coarse": gpu_util.module_from(spec_c, orc.make_field_params(spec_c, 3, regime))}
    spec_f = orc.FieldSpec("fine", encode_appearance=fine in ("a", "at"), encode_transient=fine == "at", beta_min=0.1)
    models["fine"] = gpu_util.module_from(spec_f, orc.make_field_params(spec_f, 4, regime))
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    if spec_f.encode_appearance:
        emb["a"] = torch.nn.Embedding(n_vocab, 48).to(gpu_util.DEV)
    if spec_f.encode_transient:
        emb["t"] = torch.nn.Embedding(n_vocab, 16).to(gpu_util.DEV)
    return models, emb


def _render(models, emb, rays, ts, S, I, test_time=False, white_back=True, **kw):
    from nerf_fl_amd import render_rays
    with torch.no_grad():
        return render_rays(models, emb, rays, ts, S, False, 0, 0.0, I, 32768, white_back, test_time, **kw)


@pytest.mark.parametrize("fine,R,S,I,test_time", [
    ("base", 4096, 64, 64, False),        # configs[1]
    ("at", 4096, 64, 64, False),          # configs[2]
    ("at", 16384, 128, 128, True),        # configs[4] shape (a chunk of it)
])
def test_invariants_full_size(fine, R, S, I, test_time):
    import gpu_util
    dev = gpu_util.DEV
    models, emb = _setup(fine)
    rays = orc.make_rays(R, 77).to(dev)
    ts = torch.randint(0, 100, (R,), device=dev)
    res = _render(models, emb, rays, ts, S, I, test_time=test_time, _field_raw=True)
    z = res.pop("_z_fine")
    res.pop("_field_raw_coarse"), res.pop("_field_raw_fine")
    for k, v in res.items():
        assert torch.isfinite(v).all(), k
    # depths: sorted, inside [near, far]
    assert (z[:, 1:] >= z[:, :-1]).all()
    assert (z >= 2.0 - 1e-5).all() and (z <= 6.0 + 1e-5).all()
    # weights are a sub-probability distribution; opacity is their sum
    for typ in ("coarse", "fine"):
        w = res[f"weights_{typ}"]
        assert (w >= 0).all()
        assert torch.allclose(w.sum(1), res[f"opacity_{typ}"], atol=2e-5)
        assert (res[f"opacity_{typ}"] <= 1 + 1e-5).all()
    # colours stay in the unit cube (white background adds exactly the missing opacity)
    assert (res["rgb_fine"] >= -1e-5).all()
    if fine == "at":
        assert torch.allclose(res["rgb_fine"], res["_rgb_fine_static"] + res["_rgb_fine_transient"], atol=1e-6)
        assert (res["beta"] >= 0.1 - 1e-6).all()
        assert (res["transient_sigmas"] >= 0).all()
    else:
        assert (res["rgb_fine"] <= 1 + 1e-5).all()
    # depth is a weighted mean of the sample depths
    assert (res["depth_fine"] <= 6.0 * res["opacity_fine"] + 1e-4).all()

    # rays are independent: a permuted batch gives bitwise permuted results, and so does splitting it
    perm = torch.randperm(R, device=dev)
    res_p = _render(models, emb, rays[perm], ts[perm], S, I, test_time=test_time)
    for k in res:
        assert torch.equal(res_p[k], res[k][perm]), k
    half = _render(models, emb, rays[: R // 2 + 3], ts[: R // 2 + 3], S, I, test_time=test_time)
    for k in res:
        assert torch.equal(half[k], res[k][: R // 2 + 3]), k


def test_training_step_full_size_linearity():
    """configs[1] train step: gradients are finite, reproducible up to atomic summation order,
    and linear in the upstream gradient."""
    import gpu_util
    from nerf_fl_amd import render_rays
    dev = gpu_util.DEV
    R, S, I = 4096, 64, 64
    models, emb = _setup("base")
    rays = orc.make_rays(R, 78).to(dev)
    ts = torch.zeros(R, dtype=torch.long, device=dev)
    g = torch.Generator(device=dev).manual_seed(5)
    inj = dict(perturb_rand=torch.rand(R, S, device=dev, generator=g), noise_coarse=torch.randn(R, S, device=dev, generator=g),
               u=torch.rand(R, I, device=dev, generator=g), noise_fine=torch.randn(R, S + I, device=dev, generator=g))
    target = torch.rand(R, 3, device=dev, generator=g)
    params = [p for m in models.values() for p in m.parameters()]

    def grads(scale):
        for p in params:
            p.grad = None
        res = render_rays(models, emb, rays, ts, S, False, 1.0, 1.0, I, 32768, True, False, **inj)
        loss = scale * sum(orc.nerfw_loss(res, target).values())
        loss.backward()
        return torch.cat([p.grad.flatten() for p in params]), float(loss.detach())

    g1, l1 = grads(1.0)
    g1b, l1b = grads(1.0)
    g2, _ = grads(2.0)
    assert torch.isfinite(g1).all() and l1 == l1b
    ref = g1.abs().max().item()
    assert ref > 0
    assert (g1 - g1b).abs().max().item() <= 1e-4 * ref        # fp32 atomics: order-dependent last bits only
    assert (g2 - 2 * g1).abs().max().item() <= 1e-3 * ref
