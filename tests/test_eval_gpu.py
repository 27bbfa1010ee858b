"""Chunked / graph-captured inference (reference eval.py:80-110) against direct render_rays."""
import pytest
import torch

from oracle import nerfw_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_graph", [False, True])
def test_batched_inference_matches_direct(use_graph):
    import gpu_util
    from nerf_fl_amd import PosEmbedding, render_rays
    from nerf_fl_amd.eval import batched_inference
    dev = gpu_util.DEV
    spec_c = orc.FieldSpec("coarse")
    spec_f = orc.FieldSpec("fine", encode_appearance=True, encode_transient=True, beta_min=0.1)
    models = {"coarse": gpu_util.module_from(spec_c, orc.make_field_params(spec_c, 3, "sharp")),
              "fine": gpu_util.module_from(spec_f, orc.make_field_params(spec_f, 4, "sharp"))}
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4),
           "a": torch.nn.Embedding(50, 48).to(dev), "t": torch.nn.Embedding(50, 16).to(dev)}
    R, chunk = 2500, 1024                       # 2 full chunks + a ragged one of 452 rays
    rays = orc.make_rays(R, 5, near=0.3, far=5.0).to(dev)
    ts = torch.randint(0, 50, (R,), device=dev)
    cache = {}
    got = batched_inference(models, emb, rays, ts, 128, 128, chunk=chunk, white_back=False, use_graph=use_graph,
                            _graph_cache=cache)
    if use_graph:                               # replay the same captured graph on a second "frame"
        got = batched_inference(models, emb, rays, ts, 128, 128, chunk=chunk, white_back=False, use_graph=True,
                                _graph_cache=cache)
        assert len(cache) == 1
    with torch.no_grad():
        exp = render_rays(models, emb, rays, ts, 128, False, 0, 0, 128, chunk, False, True)
    assert list(got.keys()) == list(exp.keys())
    for k in exp:
        assert got[k].shape == exp[k].shape
        assert torch.equal(got[k], exp[k]), k


def test_frame_rays_match_ray_utils_restatement():
    """nfl_gen_rays against the torch restatement of datasets/ray_utils.py (nerf_fl_amd.poses, itself checked on CPU in
    test_poses_cpu.py): whole frame and a ragged pixel range."""
    import math

    from nerf_fl_amd.eval import frame_rays
    from nerf_fl_amd.poses import get_ray_directions, get_rays, make_c2w
    H, W = 37, 29
    focal = 0.5 * W / math.tan(0.5 * 0.6911)
    K = torch.tensor([[focal, 0, W / 2], [0, focal * 1.03, H / 2 - 0.25], [0, 0, 1]], dtype=torch.float32)
    c2w = make_c2w(torch.tensor([0.3, -0.5, 0.2]), torch.tensor([0.4, -1.1, 3.7]))[:3]
    d = get_ray_directions(H, W, K).reshape(-1, 3)
    o_ref, d_ref = get_rays(d, c2w)
    exp = torch.cat([o_ref, d_ref, torch.full((H * W, 1), 2.0), torch.full((H * W, 1), 6.0)], 1)
    got = frame_rays(c2w, K, H, W, 2.0, 6.0, "cuda:0").cpu()
    assert got.shape == exp.shape
    assert (got - exp).abs().max().item() <= 2e-7
    assert torch.equal(got[:, 6:], exp[:, 6:]) and torch.equal(got[:, :3], exp[:, :3])
    part = frame_rays(c2w, K, H, W, 2.0, 6.0, "cuda:0", start=101, count=333).cpu()
    assert torch.equal(part, got[101:434])
    assert frame_rays(c2w, K, H, W, 2.0, 6.0, "cuda:0", start=5, count=0).shape == (0, 8)
