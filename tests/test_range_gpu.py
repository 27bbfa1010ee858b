"""Numerical range of the fused MLP (VERDICT r1, weak #2d).  The kernels multiply fp16 operands (split hi + lo): an
activation or weight beyond fp16's range (|x| > 65504) cannot be represented (hi = inf, lo = -inf), where the fp32
reference simply carries it.  On gfx950 the matrix cores turn inf - inf and inf * 0 into a NEGATIVE quiet NaN
(profiles/tools/fp16_probe.hip), which the next layer's relu -- one integer max on the bit pattern -- maps to zero: the
outputs of an out-of-range network are finite and look plausible.  The contract ('Numerical range'):
  * inside the range the 1e-4 parity bar holds, also for activations in the thousands;
  * outside it the kernels REPORT it: every epilogue tracks the largest fp16 operand it forms, the weight packer
    checks the weights, NFL_STATUS_RANGE is OR-ed into a device status word, and `check_finite=True` /
    `nerf_fl_amd.check_status()` raise FloatingPointError.
relu is positively homogeneous, so scaling layer 1 (weight and bias) by K and layer 2's weight by 1/K leaves the
field's function unchanged while multiplying the layer-1 activations by K: K picks the regime."""
import pytest
import torch

from oracle import nerfw_oracle as orc

pytestmark = pytest.mark.gpu


def _scaled_params(spec, seed, K):
    P = orc.make_field_params(spec, seed, "sharp")
    P["xyz_encoding_1.0.weight"] = P["xyz_encoding_1.0.weight"] * K
    P["xyz_encoding_1.0.bias"] = P["xyz_encoding_1.0.bias"] * K
    P["xyz_encoding_2.0.weight"] = P["xyz_encoding_2.0.weight"] / K
    return P


def _run(K, **extra):
    import gpu_util
    from nerf_fl_amd import PosEmbedding, render_rays
    dev = gpu_util.DEV
    spec_c, spec_f = orc.FieldSpec("coarse"), orc.FieldSpec("fine")
    P_c, P_f = _scaled_params(spec_c, 81, K), _scaled_params(spec_f, 82, K)
    models = {"coarse": gpu_util.module_from(spec_c, P_c), "fine": gpu_util.module_from(spec_f, P_f)}
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    rays = orc.make_rays(96, 83)
    with torch.no_grad():
        exp = orc.render_rays(spec_c, P_c, spec_f, P_f, rays, n_samples=64, n_importance=64, noise_std=0.0, white_back=True)
        got = render_rays(models, emb, rays.to(dev), torch.zeros(96, dtype=torch.long, device=dev), 64, False, 0, 0.0, 64,
                          32768, True, False, **extra)
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in got.items()}, exp


@pytest.mark.parametrize("K", [64.0, 1024.0])
def test_large_activations_inside_the_range_keep_parity(K):
    """Layer-1 activations up to ~K * 3 (tens to thousands): still <= 1e-4 on every output, and the status word stays clear."""
    import nerf_fl_amd
    got, exp = _run(K, check_finite=True)
    for k in exp:
        assert (got[k] - exp[k]).abs().max().item() <= 1e-4, k
    nerf_fl_amd.check_status()


def test_overflow_is_reported_not_silent():
    import nerf_fl_amd
    K = 2.0 ** 17                      # layer-1 activations ~ 1e5 .. 4e5: beyond fp16
    with pytest.raises(FloatingPointError):
        _run(K, check_finite=True)
    got, exp = _run(K)                 # without the check the call returns (wrong numbers) ...
    assert all(torch.isfinite(v).all() for v in exp.values()), "the fp32 reference carries these magnitudes"
    assert max((got[k] - exp[k]).abs().max().item() for k in exp) > 1e-4, "this input is meant to break the fp16 path"
    with pytest.raises(FloatingPointError, match="fp16's range"):
        nerf_fl_amd.check_status()     # ... and the status word has recorded why
    nerf_fl_amd.check_status()         # cleared by the raise


def test_weight_beyond_fp16_is_reported():
    import gpu_util
    import nerf_fl_amd
    from nerf_fl_amd import PosEmbedding, render_rays
    dev = gpu_util.DEV
    spec = orc.FieldSpec("coarse")
    P = orc.make_field_params(spec, 84, "default")
    P["xyz_encoding_3.0.weight"][5, 7] = 1.0e5
    models = {"coarse": gpu_util.module_from(spec, P)}
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    rays = orc.make_rays(8, 85).to(dev)
    with torch.no_grad(), pytest.raises(FloatingPointError):
        render_rays(models, emb, rays, torch.zeros(8, dtype=torch.long, device=dev), 32, False, 0, 0.0, 0, 32768, True,
                    False, check_finite=True)
