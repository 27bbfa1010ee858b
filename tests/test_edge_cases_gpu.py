"""Edge cases: empty and tiny batches, minimal and odd sample counts, against the CPU oracle."""
import pytest
import torch

from oracle import nerfw_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,S,I", [(0, 8, 4), (1, 3, 1), (5, 8, 0), (3, 33, 31), (2, 1, 0), (7, 40, 5)])
def test_small_and_empty(R, S, I):
    import gpu_util
    from nerf_fl_amd import PosEmbedding, render_rays
    dev = gpu_util.DEV
    spec_c = orc.FieldSpec("coarse")
    spec_f = orc.FieldSpec("fine", encode_appearance=True, encode_transient=True, beta_min=0.1)
    P_c, P_f = orc.make_field_params(spec_c, 61, "sharp"), orc.make_field_params(spec_f, 62, "sharp")
    models = {"coarse": gpu_util.module_from(spec_c, P_c), "fine": gpu_util.module_from(spec_f, P_f)}
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    g = torch.Generator().manual_seed(R * 100 + S)
    rays = orc.make_rays(max(R, 1), 63)[:R]
    a_emb, t_emb = torch.randn(R, 48, generator=g), torch.randn(R, 16, generator=g)
    with torch.no_grad():
        got = render_rays(models, emb, rays.to(dev), torch.zeros(R, dtype=torch.long, device=dev), S, False, 0, 0.0, I,
                          32768, True, False, a_embedded=a_emb.to(dev), t_embedded=t_emb.to(dev))
        exp = orc.render_rays(spec_c, P_c, spec_f if I > 0 else None, P_f if I > 0 else None, rays, n_samples=S,
                              n_importance=I, noise_std=0.0, white_back=True, a_emb=a_emb, t_emb=t_emb)
    assert list(got.keys()) == list(exp.keys())
    for k in exp:
        assert tuple(got[k].shape) == tuple(exp[k].shape), k
        if R:
            assert (got[k].cpu() - exp[k]).abs().max().item() <= 1e-4, k


def test_argument_errors():
    import gpu_util
    from nerf_fl_amd import NeRF, PosEmbedding, render_rays
    dev = gpu_util.DEV
    models = {"coarse": NeRF("coarse").to(dev)}
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    rays = orc.make_rays(4, 1).to(dev)
    ts = torch.zeros(4, dtype=torch.long, device=dev)
    with torch.no_grad():
        with pytest.raises(KeyError):                      # fine model missing, as in the reference
            render_rays(models, emb, rays, ts, 8, False, 0, 0, 4)
        with pytest.raises(TypeError):
            render_rays(models, emb, rays.double(), ts, 8)
        with pytest.raises(ValueError):
            render_rays(models, emb, rays[:, :6], ts, 8)
        with pytest.raises(ValueError):                    # embedding width does not match the model
            render_rays(models, {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}, rays, ts, 8)


def test_parameters_modified_between_forward_and_backward():
    """The hand-written backward reads the weights again (dgrad stream; the gradients around xyz_encoding_final are
    composed from the fp32 weights): like autograd's saved-tensor version check it refuses weights that changed."""
    import gpu_util
    from nerf_fl_amd import NeRF, PosEmbedding, render_rays
    dev = gpu_util.DEV
    models = {"coarse": NeRF("coarse").to(dev), "fine": NeRF("fine").to(dev)}
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    rays = orc.make_rays(8, 1).to(dev)
    ts = torch.zeros(8, dtype=torch.long, device=dev)
    res = render_rays(models, emb, rays, ts, 8, False, 0, 0, 8)
    loss = res["rgb_fine"].sum() + res["rgb_coarse"].sum()
    with torch.no_grad():
        models["fine"].xyz_encoding_final.weight.mul_(1.5)
    with pytest.raises(RuntimeError, match="modified in place"):
        loss.backward()
    res = render_rays(models, emb, rays, ts, 8, False, 0, 0, 8)         # an untouched pair of passes still works
    (res["rgb_fine"].sum() + res["rgb_coarse"].sum()).backward()
    assert models["fine"].xyz_encoding_final.weight.grad is not None
