"""NerfWLoss fused into the render kernels' per-ray epilogue (reference losses.py:35-50): the fused
total, its four terms and every gradient it produces against the unfused composition -- the same render_rays call
followed by the reference's loss formulas on the result dict and autograd -- on the base and the NeRF-W configuration,
with the total scaled by an arbitrary factor before backward (exercises the device-side upstream gradient)."""
import pytest
import torch

import golden_util as gu
from oracle import nerfw_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["g11_grad_cfg2", "g11_grad_cfg3_ts", "g12_stoch_grad", "g15_photo_stoch"])
def test_fused_loss_matches_unfused(name):
    import gpu_util
    from nerf_fl_amd import render_rays
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

    def grads(fused):
        for p in params:
            p.grad = None
        if fused:
            res = render_rays(*args, loss_target=target, **extra)
            total, terms = res["_nerfw_loss"], res["_nerfw_terms"]
            assert not res["rgb_fine"].requires_grad and total.requires_grad
        else:
            res = render_rays(*args, **extra)
            t = orc.nerfw_loss(res, target)
            total = sum(t.values())
            terms = torch.stack([t.get(k, torch.zeros((), device=dev)) for k in ("c_l", "f_l", "b_l", "s_l")])
        (2.5 * total).backward()
        return float(total), terms.detach().cpu(), torch.cat([p.grad.flatten() for p in params]).cpu()

    l_f, t_f, g_f = grads(True)
    l_u, t_u, g_u = grads(False)
    assert abs(l_f - l_u) <= 2e-6 * max(1.0, abs(l_u))
    assert abs(l_f - a["loss"].item()) <= 1e-4 * max(1.0, abs(a["loss"].item()))        # and the reference's own total
    assert (t_f - t_u).abs().max().item() <= 2e-6 * max(1.0, t_u.abs().max().item())
    ref = g_u.abs().max().item()
    assert ref > 0
    # same kernels, same seeds up to fp32 rounding: what is left is the atomics' summation order and the loss scale
    assert (g_f - g_u).abs().max().item() <= 2e-3 * ref
    assert abs(g_f.norm().item() - g_u.norm().item()) <= 1e-3 * g_u.norm().item()


def test_fused_loss_needs_training_mode():
    import gpu_util
    from nerf_fl_amd import NeRF, PosEmbedding, render_rays
    dev = gpu_util.DEV
    models = {"coarse": NeRF("coarse").to(dev)}
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    rays = orc.make_rays(8, 1).to(dev)
    with torch.no_grad(), pytest.raises(RuntimeError, match="TRAINING"):
        render_rays(models, emb, rays, torch.zeros(8, dtype=torch.long, device=dev), 16, False, 0, 0, 0, 32768, True, False,
                    loss_target=torch.rand(8, 3, device=dev))
