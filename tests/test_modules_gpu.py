"""GPU parity of the module-level entry points the reference also exposes on their own:
PosEmbedding / BarfPosEmbedding.forward (models/nerf.py:19-32, 61-77) and NeRF.forward on encoded
inputs (models/nerf.py:153-212), through nfl_posenc / nfl_field_forward, against the golden vectors
the real reference produced (g1_posenc, g2_field_*) and against the oracle at ragged sizes."""
import pytest
import torch

import golden_util as gu
import nerf_fl_amd
from gpu_util import DEV, module_from
from nerf_fl_amd import BarfPosEmbedding, PosEmbedding
from oracle import nerfw_oracle as orc

pytestmark = pytest.mark.gpu

PE_TOL = 1e-6        # fp32 sin/cos of |x| <= ~4 scaled by up to 2^14: the reference's own libm error is ~1e-7
TOL = 1e-4           # north_star tolerance (relative to max(1, |y|) for the unbounded softplus outputs)


def test_g1_posenc_golden():
    cfg, a = gu.load("g1_posenc")
    for n in cfg["n_freqs"]:
        got = PosEmbedding(n - 1, n)(a["x"].to(DEV)).cpu()
        exp = a[f"out_{n}"]
        assert got.shape == exp.shape
        assert torch.equal(got[:, :3], exp[:, :3])
        assert (got - exp).abs().max().item() <= PE_TOL


@pytest.mark.parametrize("epoch", [2, 5, 6, 9])
def test_barf_posenc_vs_oracle(epoch):
    x = torch.randn(3, 37, 3, generator=torch.Generator().manual_seed(epoch)) * 2
    emb = BarfPosEmbedding(9, 10, 4, 8)
    got = emb(x.to(DEV), epoch).cpu()
    exp = orc.posenc(x, 10, orc.barf_weights(10, epoch))
    assert got.shape == (3, 37, 63)
    assert (got - exp).abs().max().item() <= PE_TOL


@pytest.mark.parametrize("precision", ["f16x3"])
@pytest.mark.parametrize("name", gu.golden_names("g2_field_"))
def test_g2_field_golden(name, precision):
    cfg, a = gu.load(name)
    spec = orc.FieldSpec(**cfg["spec"])
    P = orc.make_field_params(spec, cfg["seed"], cfg["regime"])
    nerf_fl_amd.set_precision(precision)
    m = module_from(spec, P)
    with torch.no_grad():
        got = m(a["x"].to(DEV), sigma_only=cfg["sigma_only"], output_transient=cfg["output_transient"]).cpu()
    exp = a["y"]
    assert got.shape == exp.shape
    err = ((got - exp).abs() / exp.abs().clamp(min=1.0)).max().item()
    assert err <= TOL, f"{name}: {err:.3e}"


@pytest.mark.parametrize("B", [1, 31, 33, 4099])
def test_field_forward_ragged(B):
    spec = orc.FieldSpec("fine", encode_appearance=True, encode_transient=True)
    P = orc.make_field_params(spec, 77, "default")
    g = torch.Generator().manual_seed(B)
    x = torch.cat([orc.posenc(torch.rand(B, 3, generator=g) * 4 - 2, 10),
                   orc.posenc(torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1), 4),
                   torch.randn(B, 48, generator=g), torch.randn(B, 16, generator=g)], 1)
    nerf_fl_amd.set_precision("f16x3")
    m = module_from(spec, P)
    for so, ot, cols in ((False, True, 9), (False, False, 4), (True, False, 1)):
        with torch.no_grad():
            got = m(x.to(DEV), sigma_only=so, output_transient=ot).cpu()
            exp = orc.field_forward_packed(spec, P, x if not so else x[:, :63], sigma_only=so, output_transient=ot)
        assert got.shape == (B, cols) == tuple(exp.shape)
        assert ((got - exp).abs() / exp.abs().clamp(min=1.0)).max().item() <= TOL


@pytest.mark.parametrize("n_xyz,n_dir,n_a,n_tau", [(6, 2, 24, 8), (12, 3, 40, 5), (3, 1, 48, 16), (15, 4, 1, 1)])
def test_field_forward_other_widths(n_xyz, n_dir, n_a, n_tau):
    """NeRF.forward on encoded inputs for other encoder / latent widths (opt.py:25-28, --N_a, --N_tau): narrower inputs run in
    the next wider kernel instantiation on zero-padded weights (nfl_plan.h); against the pinned oracle."""
    spec = orc.FieldSpec("fine", n_emb_xyz=n_xyz, n_emb_dir=n_dir, encode_appearance=True, n_a=n_a, encode_transient=True, n_tau=n_tau)
    P = orc.make_field_params(spec, 91, "default")
    B = 77
    g = torch.Generator().manual_seed(n_xyz * 100 + n_a)
    x = torch.cat([orc.posenc(torch.rand(B, 3, generator=g) * 4 - 2, n_xyz),
                   orc.posenc(torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1), n_dir),
                   torch.randn(B, n_a, generator=g), torch.randn(B, n_tau, generator=g)], 1)
    nerf_fl_amd.set_precision("f16x3")
    m = module_from(spec, P)
    cx = 6 * n_xyz + 3
    for so, ot, cols in ((False, True, 9), (False, False, 4), (True, False, 1)):
        with torch.no_grad():
            got = m(x.to(DEV), sigma_only=so, output_transient=ot).cpu()
            exp = orc.field_forward_packed(spec, P, x if not so else x[:, :cx], sigma_only=so, output_transient=ot)
        assert got.shape == (B, cols) == tuple(exp.shape)
        assert ((got - exp).abs() / exp.abs().clamp(min=1.0)).max().item() <= TOL


def test_module_forward_rejects_cpu_and_grad():
    m = module_from(orc.FieldSpec("coarse"), orc.make_field_params(orc.FieldSpec("coarse"), 1, "default"))
    with pytest.raises(RuntimeError):
        m(torch.zeros(4, 90))
    with pytest.raises(RuntimeError):
        PosEmbedding(9, 10)(torch.zeros(4, 3))
    with pytest.raises(RuntimeError):
        m(torch.zeros(4, 90, device=DEV, requires_grad=True))
    with pytest.raises(ValueError):          # the reference raises too (no transient_encoding on this model)
        m(torch.zeros(4, 90, device=DEV))
    assert m(torch.zeros(0, 90, device=DEV), output_transient=False).shape == (0, 4)
