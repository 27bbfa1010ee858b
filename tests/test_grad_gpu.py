"""GPU gradient parity: the hand-written HIP backward (composite -> dgrad -> wgrad) against the gradients autograd
produced through the REAL reference (golden vectors g11_* / g12_stoch_grad / g14_* / g15_* / g16_* / g17_*), for
loss = sum(NerfWLoss).

Arithmetic under test: the forward is fp32-class (f16x3); the MLP part of the backward multiplies in fp16 (one
product, fp32 accumulation) on fp16-stashed activations and loss-scaled fp16 gradients, so its
error is a few 2^-11 relative per product, averaged over the samples of the batch.  Five measures per tensor, each with
its own threshold (about 2x the largest value measured over all cases, tests/report_grads.py prints them):

  max   max|g - ref| / max|ref|              every element, on the scale of the tensor
  l2    ||g - ref|| / ||ref||                every element, weighted by its share of the tensor: a population of small
                                             but wrong entries shows here even when each is below `max`
  elem  max|g - ref| / |ref|                 over elements with |ref| >= 0.1 max|ref|.  A weight-gradient entry is a sum
                                             over up to 10^5 samples of fp16-rounded terms of both signs, so its
                                             ABSOLUTE error scales with the tensor (`max`), not with the entry: a cut
                                             at 1e-3 max|ref| would only restate `max` / 1e-3
  norm  | ||g|| - ||ref|| | / ||ref||        whole tensor
  proj  |<g, r> - <ref, r>| / ||ref||        a fixed random direction r (N(0,1), seed 99): the WHOLE of a big tensor,
                                             of which the fixture otherwise stores only four rows
"""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import nerfw_oracle as orc

pytestmark = pytest.mark.gpu
CASES = ["g11_grad_cfg1", "g11_grad_cfg2", "g11_grad_cfg3", "g11_grad_cfg3_ts", "g12_stoch_grad",
         "g11_grad_rays", "g14_barf_e6", "g14_barf_e9",      # the last three also check d/d rays (learnable poses)
         "g15_photo_grad", "g15_photo_stoch",                # configs[3]: N_vocab 1500 tables, per-ray near/far, R 1024
         "g16_view_dir",                                     # view_dir kwarg
         "g16_view_dir_rays",                                # view_dir AND d/d rays (the direction encoding is then data)
         "g18_na24_tau8", "g18_na40_tau5_emb",               # other latent widths (opt.py --N_a / --N_tau), tables and kwargs
         "g18_emb6_2", "g18_emb12_4", "g18_emb3_1_barf",     # other encoder widths (opt.py:25-28): xyz 6 / dir 2, 12 / 4 (+ rays), 3 / 1 (BARF + rays)
         "g17_trained_cfg2", "g17_trained_cfg3", "g17_trained_cfg2_stoch"]   # weights after 400 reference Adam steps

# thresholds: measure -> (default, {tensor-name substring: override})
# Big tensors are pinned by their first four rows, first four columns, norm and two independent random projections
# (gradrows / gradcols / gradnorm / gradproj / gradproj2 of tests/golden/make_golden.py), small ones entirely.
# Both single-image modes round their fp16 gradients STOCHASTICALLY since round 3 (nfl_dgrad.hip: the error of a gradient is
# zero-mean and independent from sample to sample -- test_stochastic_rounding_is_unbiased below -- instead of a fixed
# function of its value), and "f16" also draws the rounding of the transposed weights its gradient chain multiplies by
# (nfl_pack.hip).  A single step's error is therefore a draw: the figures are the worst over all cases and rounding seeds
# 0..5 (MI355X, tests/report_grads.py --seed N).
#   f16w (exact weights in the chain):  max <= 2.9e-3, l2 <= 2.8e-3, elem <= 2.0e-2, norm <= 1.0e-3, proj <= 3.7e-3
#   f16  (default):                     max <= 6.0e-3, l2 <= 3.9e-3, elem <= 2.9e-2, norm <= 1.5e-3, proj <= 3.4e-3
# -- except the density heads (static_sigma / transient_sigma: one row, the bias a single number): their gradient is a sum
# over all samples of terms of both signs that largely cancel on peaky densities, so the fp16 rounding of the terms (2^-11
# each) shows relative to the much smaller sum: max / l2 / norm up to 7.7e-3, elem (the weight row) up to 3.8e-2.
# The price of the zero mean is a larger single draw than round-to-nearest gave (round 2: max <= 2.8e-3 in "f16"); what it
# buys is a training curve without the systematic offset (tests/test_psnr_parity_gpu.py).
_SIGMA = {"static_sigma": 1.5e-2, "transient_sigma": 1.5e-2}
_SIGMA_ELEM = {"static_sigma": 6e-2, "transient_sigma": 6e-2}
THRESH_F16W = {
    "max": (5e-3, _SIGMA),
    "l2": (4e-3, _SIGMA),
    "elem": (3.5e-2, _SIGMA_ELEM),
    "norm": (2e-3, _SIGMA),
    "proj": (5e-3, {}),
}
THRESH = {
    "max": (8e-3, _SIGMA),
    "l2": (6e-3, _SIGMA),
    "elem": (4.5e-2, _SIGMA_ELEM),
    "norm": (2.5e-3, _SIGMA),
    "proj": (5e-3, {}),
}


# The opt-in three-product backward (set_precision(backward="f16x3"): hi + lo weight fragments, activations and gradients,
# the forward's arithmetic; the reference's fp32 precision class).  Measured over all cases (MI355X, round 3): max <= 7.4e-4,
# l2 <= 1.2e-3, elem <= 4.8e-3, norm <= 4.6e-5, proj <= 4.5e-4 -- and on the trained-weight fixtures (g17_*), where the
# loss is smooth in the weights, max <= 1.7e-4 (one head; every other tensor ~1e-6).  What is left on the sharpened random
# fields is not arithmetic of the backward: the reference's own fp32 autograd differs from the same algorithm in fp64 by
# 7e-4 .. 1e-2 of max|g| on these fixtures (relu / density-threshold decisions that flip with the forward's last bits:
# tests/report_grads.py --floor), so 1e-4-class agreement is only defined where the reference is that well conditioned.
THRESH_X3 = {
    "max": (1.5e-3, {"gradcols": 2.5e-3}),      # four-column slices are normalised by their own (smaller) maximum: measured 1.3e-3
    "l2": (2.5e-3, {}),
    "elem": (1e-2, {}),
    "norm": (1e-4, {}),
    "proj": (1e-3, {}),
}
THRESH_X3_TRAINED = {"max": (4e-4, {}), "l2": (3e-4, {}), "elem": (2.5e-3, {}), "norm": (8e-5, {}), "proj": (5e-5, {})}


def _limit(measure, key, table=None):
    default, over = (table or THRESH)[measure]
    for sub, v in over.items():
        if sub in key:
            return v
    return default


def run_case(name, backward="f16"):
    import nerf_fl_amd
    nerf_fl_amd.set_precision("f16x3", backward=backward)
    try:
        return _run_case(name)
    finally:
        nerf_fl_amd.set_precision("f16x3", backward="f16")


def _run_case(name):
    import gpu_util
    from nerf_fl_amd import render_rays
    cfg, a = gu.load(name)
    (spec_c, P_c, spec_f, P_f), kw = gu.oracle_kwargs(cfg, a)
    dev = gpu_util.DEV
    barf = cfg.get("barf_epoch") is not None
    models = {"coarse": gpu_util.module_from(spec_c, P_c, barf)}
    if spec_f is not None:
        models["fine"] = gpu_util.module_from(spec_f, P_f, barf)
    emb = gpu_util.make_embeddings(spec_c.n_emb_xyz, barf, spec_c.n_emb_dir)
    extra, leaves = {}, {}
    if barf:
        extra["current_epoch"] = cfg["barf_epoch"]
    for k in ("perturb_rand", "noise_coarse", "u", "noise_fine"):
        if kw.get(k) is not None:
            extra[k] = kw[k].to(dev)
    if kw.get("view_dir") is not None:
        extra["view_dir"] = kw["view_dir"].to(dev)
    if "z_fine" in a:
        # the fine depths of the reference run: no gradient flows through the sampler (rendering.py:269 detaches), and
        # its discontinuities (golden_util.sampling_conditioning) would otherwise put a different loss under test
        extra["z_fine"] = a["z_fine"].to(dev)
    ts = a["ts"].to(dev)
    if cfg["kwargs_mode"] == "embedded":
        for k, kk in (("a_emb", "a_embedded"), ("t_emb", "t_embedded")):
            if kw.get(k) is not None:
                leaves[k] = kw[k].to(dev).requires_grad_(True)
                extra[kk] = leaves[k]
    else:
        for k, dim in (("a", cfg.get("n_a", 48)), ("t", cfg.get("n_tau", 16))):
            if kw.get(k + "_emb") is not None:
                table = gu.embedding_table(cfg, k)
                e = torch.nn.Embedding(table.shape[0], dim).to(dev)
                e.weight.data.copy_(table)
                emb[k] = e
                leaves["table_" + k] = e.weight
    rays = a["rays"].to(dev)
    if "grad.rays" in a:
        rays.requires_grad_(True)
        leaves["rays"] = rays
    res = render_rays(models, emb, rays, ts, cfg["S"], cfg["use_disp"], cfg["perturb"], cfg["noise_std"],
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


def _elem_rel(g, ref):
    big = ref.abs() >= 0.1 * ref.abs().max()
    return ((g - ref).abs()[big] / ref.abs()[big]).max().item() if big.any() else 0.0


def _l2(g, ref):
    return (g - ref).double().norm().item() / ref.double().norm().item()


def compare(cfg, a, got):
    """yield (measure, key, value): the four measures of the module docstring for every tensor the fixture holds"""
    for key, exp in a.items():
        if key == "grad.rays":      # origin and direction columns; near/far carry no gradient here (they are data)
            g, ref = got["rays"][:, :6], exp[:, :6]
            yield "max", key, (g - ref).abs().max().item() / ref.abs().max().item()
            yield "l2", key, _l2(g, ref)
            yield "elem", key, _elem_rel(g, ref)
            yield "norm", key, abs(g.norm().item() - ref.norm().item()) / ref.norm().item()
        elif key.startswith("grad."):
            g = got[key[5:]]
            if exp.abs().max().item() == 0.0:
                assert g.abs().max().item() == 0.0, key
                continue
            yield "max", key, (g - exp).abs().max().item() / exp.abs().max().item()
            yield "l2", key, _l2(g, exp)
            yield "elem", key, _elem_rel(g, exp)
            yield "norm", key, abs(g.norm().item() - exp.norm().item()) / exp.norm().item()
        elif key.startswith("gradrows."):
            g = got[key[9:]][:4]
            yield "max", key, (g - exp).abs().max().item() / exp.abs().max().item()
            yield "l2", key, _l2(g, exp)
            yield "elem", key, _elem_rel(g, exp)
        elif key.startswith("gradcols."):       # the first four input-feature columns (the rows above are output features)
            g = got[key[9:]][:, :4]
            yield "max", key, (g - exp).abs().max().item() / max(exp.abs().max().item(), 1e-30)
            yield "l2", key, _l2(g, exp)
        elif key.startswith("gradnorm."):
            yield "norm", key, abs(got[key[9:]].norm().item() - exp.item()) / exp.item()
        elif key.startswith("gradproj.") or key.startswith("gradproj2."):       # two independent random projections (seeds 99, 100)
            name, seed = (key[9:], 99) if key.startswith("gradproj.") else (key[10:], 100)
            g = got[name]
            pr = torch.from_numpy(np.random.default_rng(seed).standard_normal(g.numel()).astype(np.float32))
            yield "proj", key, abs(float((g.flatten().double() * pr.double()).sum()) - exp.item()) / a["gradnorm." + name].item()


@pytest.mark.parametrize("backward", ["f16", "f16w", "f16x3"])
@pytest.mark.parametrize("name", CASES)
def test_gradients_vs_reference(name, backward):
    cfg, a, got, loss = run_case(name, backward)
    assert abs(loss - a["loss"].item()) <= 1e-4 * max(1.0, abs(a["loss"].item()))
    table = {"f16": THRESH, "f16w": THRESH_F16W}.get(backward) or (THRESH_X3_TRAINED if name.startswith("g17_") else THRESH_X3)
    bad, seen = {}, {m: 0 for m in THRESH}
    for measure, key, val in compare(cfg, a, got):
        seen[measure] += 1
        if not val <= _limit(measure, key, table):
            bad[(measure, key)] = (val, _limit(measure, key, table))
    assert seen["max"] > 10 and seen["elem"] > 10 and seen["norm"] > 10
    if any(k.startswith("gradproj.") for k in a):
        assert seen["proj"] >= 5
    assert not bad, f"{name}: {bad}"


def test_stochastic_rounding_is_unbiased():
    """The fp16 rounding of the gradients is drawn (v_cvt_sr_f16_f32, seeded): averaged over 16 seeds the error of every
    trunk bias gradient -- a plain sum of the chain's gradients over all samples -- must shrink like independent zero-mean
    noise does (1/sqrt(16) = 0.25; measured 0.15 .. 0.29), which a rounding that is a function of the value (round to
    nearest: the same error under every seed, ratio 1) or a biased draw would not.  In "f16w", whose chain multiplies by the
    exact weights: in "f16" the rounding of the transposed weights is part of the error and is a function of the weight bits,
    not of the seed (it is redrawn when the optimizer moves the weights)."""
    import nerf_fl_amd
    n, draws = 16, []
    try:
        for seed in range(n):
            nerf_fl_amd.set_rounding_seed(seed)
            cfg, a, got, _ = run_case("g11_grad_cfg2", "f16w")
            draws.append(got)
    finally:
        nerf_fl_amd.set_rounding_seed(0)
    keys = [k[5:] for k in a if k.startswith("grad.") and "xyz_encoding" in k and k.endswith(".bias")]
    assert len(keys) >= 16
    ratios = {}
    for k in keys:
        ref = a["grad." + k].double()
        single = sum(((d[k].double() - ref).norm() / ref.norm()).item() for d in draws) / n
        mean = ((sum(d[k].double() for d in draws) / n - ref).norm() / ref.norm()).item()
        assert single > 1e-4, "the draws carry no rounding error to average: the test would be vacuous"
        ratios[k] = mean / single
    print("error of the mean over 16 rounding seeds / error of one draw:", " ".join(f"{v:.2f}" for v in ratios.values()))
    assert max(ratios.values()) <= 0.4, ratios


@pytest.mark.parametrize("backward", ["f16", "f16x3"])
def test_weight_gradients_are_bit_reproducible(backward):
    """The weight / bias gradients are sums over ~1e5 samples per element: the streaming kernel's workgroups store partial sums and a
    reduction adds them in a fixed order (no atomics), and the rounding draws are a function of the seed and the data -- so two
    backward passes over the same inputs must agree to the last bit.  (The latent-table and ray gradients are still scattered
    with fp32 atomics in dgrad: not asserted.)"""
    _, _, g1, _ = run_case("g11_grad_cfg3_ts", backward)
    _, _, g2, _ = run_case("g11_grad_cfg3_ts", backward)
    keys = [k for k in g1 if k.startswith("coarse.") or k.startswith("fine.")]
    assert len(keys) > 40
    diff = [k for k in keys if not torch.equal(g1[k], g2[k])]
    assert not diff, diff
