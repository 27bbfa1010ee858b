"""GPU parity: the HIP render_rays against (a) the golden vectors the real
reference produced and (b) the CPU oracle, on identical inputs and injected
randomness.  Tolerance: 1e-4 abs on every output (BASELINE.json north_star)."""
import pytest
import torch

import golden_util as gu
from oracle import nerfw_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-4
RENDER = [n for n in gu.golden_names("g") if n[:2] in ("g4", "g5", "g6", "g7", "g8", "g9")
          or n.startswith(("g10", "g12_stoch_base", "g12_stoch_nerfw", "g13", "g14_barf_e2", "g15", "g16", "g17", "g18"))]


# Importance sampling (rendering.py:7-46) is discontinuous / ill-conditioned in the coarse weights: see
# golden_util.sampling_conditioning.  With trained-like weights (empty space behind the surface, opacity ~ 1) a coarse
# weight that differs by 1e-7 -- three orders inside the parity bar -- moves some fine depths by 1e-4..1e-3 (the u = 1
# draw flips between the last bin and the clamp; a draw in a bin of probability 2e-5 moves by 1e-7 / 2e-5 of a bin), and
# a trained field is steep enough (sin(512 z) features) for that to change a per-sample output by 1e-1.  The reference
# differs from ITSELF by as much between two devices.  Three comparisons per fixture therefore:
#   * test_render_at_reference_depths: the fine depths of the reference run (fixture array `z_fine`) are injected, so
#     field + compositing are held to 1e-4 on EVERY ray and EVERY output, stochastic fixtures included, no allowance;
#   * tests/test_sample_pdf_gpu.py: on identical inputs the HIP sampler's draws are bit-identical to the reference's;
#   * test_render_vs_golden (here): end to end through the HIP sampler.  The coarse pass: 1e-4 everywhere.  Every fine
#     depth must lie within what the reference's own conditioning allows for coarse weights agreeing to 1e-7 absolute
#     (golden_util.sampling_conditioning); on rays whose depths reproduce the reference's to Z_SAME every output must
#     meet 1e-4; on the others the per-ray outputs must agree to LOOSE (per-sample arrays shift by one position when a
#     draw changes bins, and are not compared index by index there).
Z_SAME = 1e-5
LOOSE = 2e-2
PER_SAMPLE = ("weights_fine", "transient_sigmas")


def _worst(got, exp, keys, rows=None):
    out = {}
    for k in keys:
        assert got[k].shape == exp[k].shape, k
        d = (got[k] - exp[k]).abs()
        if rows is not None:
            d = d[rows]
        out[k] = d.max().item() if d.numel() else 0.0
    return out


@pytest.mark.parametrize("name", RENDER)
def test_render_vs_golden(name):
    import gpu_util
    cfg, a = gu.load(name)
    specs, kw = gu.oracle_kwargs(cfg, a)
    got = gpu_util.hip_render(specs, a["rays"], kw, precision="f16x3", field_raw=cfg["I"] > 0)
    z_hip = got.pop("_z_fine", None)
    for k in ("_field_raw_coarse", "_field_raw_fine"):
        got.pop(k, None)
    assert list(got.keys()) == cfg["keys"], "result keys / order must match the reference"
    exp = {k: a["out." + k] for k in cfg["keys"]}
    coarse_keys = [k for k in cfg["keys"] if k.endswith("_coarse")]
    fine_keys = [k for k in cfg["keys"] if k not in coarse_keys]
    worst = _worst(got, exp, coarse_keys)                      # the coarse pass does not depend on the sampler
    R = a["rays"].shape[0]
    same = torch.ones(R, dtype=torch.bool)
    if z_hip is not None:
        dz = (z_hip - a["z_fine"]).abs().max(1)[0].double()
        same = dz <= Z_SAME
        bound = gu.fixture_conditioning(cfg, a)
        unexplained = (dz > bound + Z_SAME).nonzero().flatten().tolist()
        assert not unexplained, (f"{name}: fine depths of rays {unexplained[:8]} differ from the reference's by more than its "
                                 f"conditioning explains: dz {dz[unexplained[:8]].tolist()} bound {bound[unexplained[:8]].tolist()}")
    # per-ray outputs: rays whose depths agree to Z_SAME.  Per-sample arrays (raw field values at the depths: a trained
    # transient density moves by 3e-4 for ONE ulp of z, 4.8e-7 at z = 5): rays whose depths are bit-identical; the
    # injected-depth test below covers them on every ray.
    worst.update(_worst(got, exp, [k for k in fine_keys if k not in PER_SAMPLE], same))
    if z_hip is not None:
        worst.update(_worst(got, exp, [k for k in fine_keys if k in PER_SAMPLE], dz == 0))
    bad = {k: v for k, v in worst.items() if not v <= TOL}
    assert not bad, f"{name}: max abs err over {TOL} (rays with the reference's fine depths): {bad} (all: {worst})"
    n_diff = int((~same).sum())
    if n_diff:
        per_ray = [k for k in fine_keys if k not in PER_SAMPLE]
        loose = _worst(got, exp, per_ray, ~same)
        print(f"{name}: the fine depths of {n_diff} of {R} rays differ from the reference's (max {dz.max().item():.2e}, all within "
              f"its conditioning); worst per-ray output difference among them {max(loose.values()):.2e}")
        bad = {k: v for k, v in loose.items() if not v <= LOOSE}
        assert not bad, f"{name}: rays with different fine depths differ by more than {LOOSE}: {bad}"


@pytest.mark.parametrize("name", [n for n in RENDER if not n.startswith("g4")])
def test_render_at_reference_depths(name):
    """Same call with the reference's fine depths injected: 1e-4 on every output of every ray."""
    import gpu_util
    cfg, a = gu.load(name)
    specs, kw = gu.oracle_kwargs(cfg, a)
    kw["z_fine"] = a["z_fine"]
    got = gpu_util.hip_render(specs, a["rays"], kw, precision="f16x3")
    assert list(got.keys()) == cfg["keys"]
    worst = _worst(got, {k: a["out." + k] for k in cfg["keys"]}, cfg["keys"])
    bad = {k: v for k, v in worst.items() if not v <= TOL}
    assert not bad, f"{name}: max abs err over {TOL}: {bad} (all: {worst})"


@pytest.mark.parametrize("name", ["g5_cfg2_base", "g6_cfg3_nerfw", "g10_cfg5_xyz15"])
def test_field_raw_vs_oracle(name):
    """Per-sample field outputs (sigma, rgb, transient heads) against the oracle's MLP
    evaluated at the depths the kernel actually used."""
    import gpu_util
    cfg, a = gu.load(name)
    specs, kw = gu.oracle_kwargs(cfg, a)
    spec_c, P_c, spec_f, P_f = specs
    got = gpu_util.hip_render(specs, a["rays"], kw, precision="f16x3", field_raw=True)
    rays = a["rays"]
    z = got["_z_fine"]
    R, F = z.shape
    xyz = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]
    enc = orc.posenc(xyz.reshape(-1, 3), spec_f.n_emb_xyz)
    side = [orc.posenc(rays[:, 3:6], 4)]
    if spec_f.encode_appearance:
        side.append(kw["a_emb"])
    dir_a = torch.cat(side, 1).repeat_interleave(F, 0)
    use_t = spec_f.encode_transient and kw["output_transient"]
    tau = kw["t_emb"].repeat_interleave(F, 0) if use_t else None
    with torch.no_grad():
        o = orc.field_forward(spec_f, P_f, enc, dir_a, tau)
    raw = got["_field_raw_fine"]
    assert (raw[:, 0:3] - o["rgb"]).abs().max().item() <= 2e-5
    rel = ((raw[:, 3] - o["sigma"]).abs() / (1 + o["sigma"].abs())).max().item()
    assert rel <= 2e-5, rel
    if use_t:
        assert (raw[:, 4:7] - o["rgb_t"]).abs().max().item() <= 2e-5
        assert ((raw[:, 7] - o["sigma_t"]).abs() / (1 + o["sigma_t"].abs())).max().item() <= 2e-5
        assert ((raw[:, 8] - o["beta"]).abs() / (1 + o["beta"].abs())).max().item() <= 2e-5


FAST_TOL = 1e-2      # the opt-in single-product mode: ~2^-11 per product, NOT the 1e-4 bar


@pytest.mark.parametrize("name", ["g5_cfg2_base_default", "g6_cfg3_nerfw", "g10_cfg5_xyz15"])
def test_fast_f16_mode_runs_and_is_close(name):
    """`set_precision("f16")` (one fp16 product per MFMA step) is a different instantiation of the fused kernel: keep
    it exercised, to the accuracy it is documented to have."""
    import gpu_util
    import nerf_fl_amd
    cfg, a = gu.load(name)
    specs, kw = gu.oracle_kwargs(cfg, a)
    try:
        got = gpu_util.hip_render(specs, a["rays"], kw, precision="f16")
    finally:
        nerf_fl_amd.set_precision("f16x3")
    assert list(got.keys()) == cfg["keys"]
    for k in cfg["keys"]:
        if k in PER_SAMPLE or k.startswith("weights_"):
            continue                    # per-sample weights move with the sampled depths; the per-ray outputs are the check
        err = (got[k] - a["out." + k]).abs().max().item()
        assert err <= FAST_TOL, (name, k, err)
