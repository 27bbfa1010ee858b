"""Pin the CPU oracle against the golden vectors produced by the real reference
(tests/golden/make_golden.py).  Forward tolerance 1e-5 abs; the
fp32-vs-fp64 noise floor of the reference itself is ~4e-6."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import nerfw_oracle as orc

TOL = 1e-5


def test_g1_posenc():
    cfg, a = gu.load("g1_posenc")
    for n in cfg["n_freqs"]:
        got = orc.posenc(a["x"], n)
        assert got.shape == a[f"out_{n}"].shape
        assert torch.equal(got, a[f"out_{n}"])


@pytest.mark.parametrize("name", gu.golden_names("g2_field_"))
def test_g2_field(name):
    cfg, a = gu.load(name)
    spec = orc.FieldSpec(**cfg["spec"])
    P = orc.make_field_params(spec, cfg["seed"], cfg["regime"])
    got = orc.field_forward_packed(spec, P, a["x"], sigma_only=cfg["sigma_only"],
                                   output_transient=cfg["output_transient"])
    assert got.shape == a["y"].shape
    assert (got - a["y"]).abs().max().item() <= TOL


def test_g3b_sample_pdf_from_coarse():
    """sample_pdf as render_rays calls it (mid-point bins of stored coarse depths, interior weights) + concat + sort."""
    cfg, a = gu.load("g3b_sample_pdf_coarse")
    I = cfg["n_importance"]
    z, w = a["z_coarse"], a["weights_coarse"]
    mids = 0.5 * (z[:, :-1] + z[:, 1:])
    for mode, u in (("det", torch.linspace(0, 1, I).expand(z.shape[0], I)), ("rnd", a["u"])):
        got = orc.sample_pdf(mids, w[:, 1:-1], u)
        assert torch.equal(got, a[mode])
        assert torch.equal(torch.sort(torch.cat([z, got], 1), 1)[0], a[f"z_fine_{mode}"])


def test_g3_sample_pdf():
    cfg, a = gu.load("g3_sample_pdf")
    I = cfg["n_importance"]
    R = a["bins"].shape[0]
    det = orc.sample_pdf(a["bins"], a["weights"], torch.linspace(0, 1, I).expand(R, I))
    assert torch.equal(det, a["det"])
    rnd = orc.sample_pdf(a["bins"], a["weights"], a["u"])
    assert torch.equal(rnd, a["rnd"])


RENDER = [n for n in gu.golden_names("g") if n[:2] in ("g4", "g5", "g6", "g7", "g8", "g9")
          or n.startswith(("g10", "g12", "g13", "g14", "g15", "g16", "g17", "g18"))]


@pytest.mark.parametrize("name", RENDER)
def test_render_forward(name):
    cfg, a = gu.load(name)
    (spec_c, P_c, spec_f, P_f), kw = gu.oracle_kwargs(cfg, a)
    kw.pop("barf_epoch", None)
    with torch.no_grad():
        res = orc.render_rays(spec_c, P_c, spec_f, P_f, a["rays"], return_z=cfg["I"] > 0, **kw)
    if cfg["I"] > 0:        # the merged, sorted fine depths the reference's fine pass used (captured by the generator)
        assert (res.pop("_z_fine") - a["z_fine"]).abs().max().item() <= 5e-6   # a few ulps of z: the sampler amplifies last-bit weight differences
    assert list(res.keys()) == cfg["keys"], "dict key order must match the reference"
    for k in cfg["keys"]:
        exp = a["out." + k]
        assert res[k].shape == exp.shape, k
        err = (res[k] - exp).abs().max().item()
        assert err <= TOL, f"{name}:{k} max abs err {err:.3e}"


GRAD = gu.golden_names("g11_") + ["g12_stoch_grad", "g14_barf_e6", "g14_barf_e9", "g15_photo_grad", "g15_photo_stoch",
                                   "g16_view_dir", "g17_trained_cfg2", "g17_trained_cfg3", "g17_trained_cfg2_stoch",
                                   "g16_view_dir_rays", "g18_emb6_2", "g18_emb12_4", "g18_emb3_1_barf",
                                   "g18_na24_tau8", "g18_na40_tau5_emb"]


@pytest.mark.parametrize("name", GRAD)
def test_render_gradients(name):
    cfg, a = gu.load(name)
    (spec_c, P_c, spec_f, P_f), kw = gu.oracle_kwargs(cfg, a)
    kw.pop("barf_epoch", None)
    rays = a["rays"].clone()
    leaves = {}
    for tag, P in (("coarse", P_c), ("fine", P_f)):
        if P is not None:
            for n, p in P.items():
                p.requires_grad_(True)
                leaves[f"{tag}.{n}"] = p
    for k in ("a_emb", "t_emb"):
        if kw[k] is not None:
            kw[k] = kw[k].clone().requires_grad_(True)
            leaves[k] = kw[k]
    if "grad.rays" in a:
        rays.requires_grad_(True)
        leaves["rays"] = rays
    res = orc.render_rays(spec_c, P_c, spec_f, P_f, rays, **kw)
    loss = sum(orc.nerfw_loss(res, a["target"]).values())
    assert abs(loss.item() - a["loss"].item()) <= 1e-5 * max(1.0, abs(a["loss"].item()))
    loss.backward()
    checked = 0
    for key, exp in a.items():
        if key.startswith("grad.") and not key.startswith("grad.table_"):
            g = leaves[key[5:]].grad
            scale = max(exp.abs().max().item(), 1e-6)
            assert (g - exp).abs().max().item() <= 2e-4 * scale + 1e-7, key
            checked += 1
        elif key.startswith("gradrows."):
            g = leaves[key[9:]].grad[:4]
            scale = max(exp.abs().max().item(), 1e-6)
            assert (g - exp).abs().max().item() <= 2e-4 * scale + 1e-7, key
            checked += 1
        elif key.startswith("gradcols."):
            g = leaves[key[9:]].grad[:, :4]
            scale = max(exp.abs().max().item(), 1e-6)
            assert (g - exp).abs().max().item() <= 2e-4 * scale + 1e-7, key
            checked += 1
        elif key.startswith("gradnorm."):
            g = leaves[key[9:]].grad
            assert abs(g.norm().item() - exp.item()) <= 2e-4 * max(exp.item(), 1e-6) + 1e-7, key
        elif key.startswith("gradproj.") or key.startswith("gradproj2."):
            name, seed = (key[9:], 99) if key.startswith("gradproj.") else (key[10:], 100)
            g = leaves[name].grad
            pr = torch.from_numpy(np.random.default_rng(seed).standard_normal(g.numel()).astype(np.float32))
            got = float((g.flatten().double() * pr.double()).sum())
            assert abs(got - exp.item()) <= 2e-4 * a["gradnorm." + name].item() + 1e-7, key
    if "grad.table_a" in a:   # scatter of the per-ray latent grads into the table
        ts = a["ts"]
        for k, dim in (("a", cfg.get("n_a", 48)), ("t", cfg.get("n_tau", 16))):
            if f"grad.table_{k}" not in a:
                continue
            tab = torch.zeros(cfg["n_vocab"], dim).index_add_(0, ts, leaves[f"{k}_emb"].grad)
            exp = a[f"grad.table_{k}"]
            assert (tab - exp).abs().max().item() <= 2e-4 * exp.abs().max().item() + 1e-7
    assert checked > 10
