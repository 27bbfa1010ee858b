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
