"""GPU gradient parity: the hand-written HIP backward (composite -> dgrad -> wgrad)
against the gradients autograd produced through the REAL reference (golden vectors
g11_*/g12_stoch_grad), for loss = sum(NerfWLoss).  The forward is fp32-class
(f16x3); the backward multiplies in bf16 (dgrad split 3x, wgrad single product over
>= 2048 samples), so the tolerance is relative: |g - g_ref| <= GTOL * max|g_ref|
per tensor."""
import pytest
import torch

import golden_util as gu
from oracle import nerfw_oracle as orc

pytestmark = pytest.mark.gpu
GTOL = 2e-2
CASES = ["g11_grad_cfg1", "g11_grad_cfg2", "g11_grad_cfg3", "g11_grad_cfg3_ts", "g12_stoch_grad"]


def run_case(name):
    import gpu_util
    import nerf_fl_amd
    from nerf_fl_amd import PosEmbedding, render_rays
    cfg, a = gu.load(name)
    (spec_c, P_c, spec_f, P_f), kw = gu.oracle_kwargs(cfg, a)
    nerf_fl_amd.set_precision("f16x3")
    dev = gpu_util.DEV
    models = {"coarse": gpu_util.module_from(spec_c, P_c)}
    if spec_f is not None:
        models["fine"] = gpu_util.module_from(spec_f, P_f)
    emb = {"xyz": PosEmbedding(spec_c.n_emb_xyz - 1, spec_c.n_emb_xyz), "dir": PosEmbedding(3, 4)}
    extra, leaves = {}, {}
    for k in ("perturb_rand", "noise_coarse", "u", "noise_fine"):
        if kw.get(k) is not None:
            extra[k] = kw[k].to(dev)
    ts = a["ts"].to(dev)
    if cfg["kwargs_mode"] == "embedded":
        for k, kk in (("a_emb", "a_embedded"), ("t_emb", "t_embedded")):
            if kw.get(k) is not None:
                leaves[k] = kw[k].to(dev).requires_grad_(True)
                extra[kk] = leaves[k]
    else:
        for k, dim, off in (("a", 48, 4), ("t", 16, 5)):
            if kw.get(k + "_emb") is not None:
                e = torch.nn.Embedding(cfg["n_vocab"], dim).to(dev)
                e.weight.data.copy_(orc.make_embedding_table(cfg["n_vocab"], dim, cfg["seed"] + off))
                emb[k] = e
                leaves["table_" + k] = e.weight
    res = render_rays(models, emb, a["rays"].to(dev), ts, cfg["S"], cfg["use_disp"], cfg["perturb"], cfg["noise_std"],
                      cfg["I"], 32768, cfg["white_back"], False, **extra)
    assert list(res.keys()) == cfg["keys"]
    loss = sum(orc.nerfw_loss(res, a["target"].to(dev)).values())
    loss.backward()
    torch.cuda.synchronize()
    got = {}
    for tag, m in models.items():
        for n, p in m.named_parameters():
            got[f"{tag}.{n}"] = p.grad.detach().cpu()
    for k, v in leaves.items():
        got[k] = v.grad.detach().cpu()
    return cfg, a, got, float(loss.detach())


def compare(cfg, a, got):
    """yield (key, max abs err, max abs ref)"""
    for key, exp in a.items():
        if key.startswith("grad.") and key != "grad.rays":
            yield key, (got[key[5:]] - exp).abs().max().item(), exp.abs().max().item()
        elif key.startswith("gradrows."):
            yield key, (got[key[9:]][:4] - exp).abs().max().item(), exp.abs().max().item()
        elif key.startswith("gradnorm."):
            yield key, abs(got[key[9:]].norm().item() - exp.item()), exp.item()


@pytest.mark.parametrize("name", CASES)
def test_gradients_vs_reference(name):
    cfg, a, got, loss = run_case(name)
    assert abs(loss - a["loss"].item()) <= 1e-4 * max(1.0, abs(a["loss"].item()))
    bad, n = {}, 0
    for key, err, ref in compare(cfg, a, got):
        n += 1
        if not err <= GTOL * ref + 1e-7:
            bad[key] = (err, ref)
    assert n > 10
    assert not bad, f"{name}: {bad}"


def test_rays_gradient_not_built():
    import gpu_util
    from nerf_fl_amd import NeRF, PosEmbedding, render_rays
    models = {"coarse": NeRF("coarse").to(gpu_util.DEV)}
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    rays = orc.make_rays(8, 1).to(gpu_util.DEV).requires_grad_(True)
    with pytest.raises(NotImplementedError):
        render_rays(models, emb, rays, torch.zeros(8, dtype=torch.long, device=gpu_util.DEV), 8)
