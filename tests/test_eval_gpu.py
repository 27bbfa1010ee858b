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


def _nerfw_models(dev, n_vocab=50):
    import gpu_util
    from nerf_fl_amd import PosEmbedding
    spec_c = orc.FieldSpec("coarse")
    spec_f = orc.FieldSpec("fine", encode_appearance=True, encode_transient=True, beta_min=0.1)
    models = {"coarse": gpu_util.module_from(spec_c, orc.make_field_params(spec_c, 3, "sharp")),
              "fine": gpu_util.module_from(spec_f, orc.make_field_params(spec_f, 4, "sharp"))}
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4),
           "a": torch.nn.Embedding(n_vocab, 48).to(dev), "t": torch.nn.Embedding(n_vocab, 16).to(dev)}
    return models, emb


def test_camera_prologue_equals_materialised_rays():
    """Rays generated inside the render kernel's prologue (CameraRays: only pose + intrinsics cross the boundary) give
    bit-identical results to rendering the ray matrix nfl_gen_rays writes for the same pixels."""
    import gpu_util
    from nerf_fl_amd import CameraRays, render_rays
    from nerf_fl_amd.eval import frame_rays, fov60_intrinsics
    from nerf_fl_amd.poses import make_c2w
    dev = gpu_util.DEV
    models, emb = _nerfw_models(dev)
    H, W = 23, 31
    K = fov60_intrinsics(W, H)
    c2w = make_c2w(torch.tensor([0.1, -0.2, 0.05]), torch.tensor([0.1, -0.2, 3.9]))[:3]
    cam = CameraRays(c2w, K, H, W, 2.0, 6.0, dev, start=17, count=600)
    rays = frame_rays(c2w, K, H, W, 2.0, 6.0, dev, start=17, count=600)
    ts = torch.randint(0, 50, (600,), device=dev)
    with torch.no_grad():
        a = render_rays(models, emb, cam, ts, 64, False, 0, 0, 64, 32768, True, True)
        b = render_rays(models, emb, rays, ts, 64, False, 0, 0, 64, 32768, True, True)
        c = render_rays(models, emb, cam.slice(100, 228), ts[100:228], 64, False, 0, 0, 64, 32768, True, True)
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert torch.equal(c[k], b[k][100:228]), k
    with pytest.raises(RuntimeError):          # inference input only
        render_rays(models, emb, cam, ts, 64, False, 0, 0, 64, 32768, True, False)


@pytest.mark.parametrize("use_graph", [False, True])
def test_render_frame_and_sharded_video(use_graph):
    """render_frame (camera prologue, device-side clip -> uint8, ts as an int / None with a_embedded) and render_video
    (frames sharded over ranks without a collective) against the plain path (reference eval.py:162-210)."""
    import gpu_util
    from nerf_fl_amd.eval import batched_inference, dolly_path, frame_rays, render_frame, render_video, fov60_intrinsics, to_uint8
    from nerf_fl_amd.poses import make_c2w
    dev = gpu_util.DEV
    models, emb = _nerfw_models(dev)
    H, W, S, I = 20, 24, 32, 32
    K = fov60_intrinsics(W, H)
    poses = dolly_path(make_c2w(torch.tensor([0.0, 0.1, 0.0]), torch.tensor([0.0, 0.0, 4.0]))[:3], n_frames=5)
    kw = dict(chunk=128, white_back=True, device=dev, output_transient=False)
    img, res = render_frame(models, emb, poses[2], K, H, W, 2.0, 6.0, S, I, ts=7, use_graph=use_graph, **kw)
    rays = frame_rays(poses[2], K, H, W, 2.0, 6.0, dev)
    exp = batched_inference(models, emb, rays, torch.full((H * W,), 7, dtype=torch.long, device=dev), S, I, chunk=128,
                            white_back=True, output_transient=False)
    assert img.dtype == torch.uint8 and tuple(img.shape) == (H, W, 3)
    assert torch.equal(res["rgb_fine"], exp["rgb_fine"])
    assert torch.equal(img.view(-1, 3), (exp["rgb_fine"].clamp(0, 1) * 255).to(torch.uint8))
    assert torch.equal(to_uint8(torch.tensor([[-0.5, 0.5, 1.7]])), torch.tensor([[0, 127, 255]], dtype=torch.uint8))
    # ts=None (reference eval.py:94) with the appearance code given explicitly, as the notebooks do
    a_emb = emb["a"](torch.tensor([7], device=dev))
    img2, _ = render_frame(models, emb, poses[2], K, H, W, 2.0, 6.0, S, I, ts=None, a_embedded=a_emb.expand(H * W, -1)
                           if not use_graph else a_emb, use_graph=use_graph, **kw)
    assert torch.equal(img2, img)
    # two ranks' shards = the whole video
    full_lo, full = render_video(models, emb, poses, K, H, W, 2.0, 6.0, S, I, ts=7, **kw)
    lo0, part0 = render_video(models, emb, poses, K, H, W, 2.0, 6.0, S, I, rank=0, world=2, ts=7, **kw)
    lo1, part1 = render_video(models, emb, poses, K, H, W, 2.0, 6.0, S, I, rank=1, world=2, ts=7, **kw)
    assert (full_lo, lo0, lo1) == (0, 0, 3) and part0.shape[0] == 3 and part1.shape[0] == 2
    assert torch.equal(torch.cat([part0, part1]), full)
    assert torch.equal(full[2], img)


def test_graph_cache_is_keyed_by_everything_that_changes_the_launches():
    """One cache, several call signatures (VERDICT r2 weak #10a): `output_transient`, the presence of a_embedded /
    view_dir and the model objects each get their own capture; the same signature replays; every result equals the
    eager call's."""
    import gpu_util
    from nerf_fl_amd import render_rays
    from nerf_fl_amd.eval import batched_inference
    dev = gpu_util.DEV
    models, emb = _nerfw_models(dev)
    models2, _ = _nerfw_models(dev)
    with torch.no_grad():
        for p in models2["fine"].parameters():
            p.mul_(1.05)
    R, chunk, S, I = 700, 256, 32, 32
    rays = orc.make_rays(R, 7, near=0.3, far=5.0).to(dev)
    ts = torch.randint(0, 50, (R,), device=dev)
    vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=1)
    a_one = emb["a"](torch.tensor([3], device=dev)).detach()
    cache = {}
    calls = [(models, {}), (models, {"output_transient": False}), (models, {"view_dir": vd}),
             (models, {"a_embedded": a_one, "output_transient": False}), (models2, {}), (models, {})]
    sizes = []
    for m, kw in calls:
        got = batched_inference(m, emb, rays, ts, S, I, chunk=chunk, white_back=True, use_graph=True, _graph_cache=cache, **kw)
        sizes.append(len(cache))
        kw_e = dict(kw)
        if "a_embedded" in kw_e:
            kw_e["a_embedded"] = kw_e["a_embedded"].expand(R, -1)
        with torch.no_grad():
            exp = render_rays(m, emb, rays, ts, S, False, 0, 0, I, chunk, True, True, **kw_e)
        assert list(got) == list(exp)
        for k in exp:
            assert torch.equal(got[k], exp[k]), (kw.keys(), k)
    assert sizes == [1, 2, 3, 4, 5, 5]


def test_graph_replay_follows_the_weights():
    """A cached capture must not render with the weights it was captured on (VERDICT r2 weak #10b): after
    load_state_dict and after an in-place optimizer-style update, with NO eager render in between, replay == eager."""
    import gpu_util
    from nerf_fl_amd import render_rays
    from nerf_fl_amd.eval import batched_inference, _GRAPH_CACHE
    dev = gpu_util.DEV
    models, emb = _nerfw_models(dev)
    R, chunk, S, I = 512, 256, 32, 32
    rays = orc.make_rays(R, 8, near=0.3, far=5.0).to(dev)
    ts = torch.randint(0, 50, (R,), device=dev)
    n0 = len(_GRAPH_CACHE)
    first = batched_inference(models, emb, rays, ts, S, I, chunk=chunk, white_back=True, use_graph=True)
    assert len(_GRAPH_CACHE) == n0 + 1                                  # the module-level default cache keeps the capture ...
    spec_f = orc.FieldSpec("fine", encode_appearance=True, encode_transient=True, beta_min=0.1)
    models["fine"].load_state_dict({k: v.to(dev) for k, v in orc.make_field_params(spec_f, 44, "sharp").items()})
    second = batched_inference(models, emb, rays, ts, S, I, chunk=chunk, white_back=True, use_graph=True)
    assert len(_GRAPH_CACHE) == n0 + 1                                  # ... and the second call replays it
    with torch.no_grad():
        exp = render_rays(models, emb, rays, ts, S, False, 0, 0, I, chunk, True, True)
    assert not torch.equal(first["rgb_fine"], second["rgb_fine"])
    for k in exp:
        assert torch.equal(second[k], exp[k]), k
    with torch.no_grad():                                                # what an optimizer step does: in place + version bump
        for p in models["coarse"].parameters():
            p.add_(0.01 * torch.randn_like(p))
    third = batched_inference(models, emb, rays, ts, S, I, chunk=chunk, white_back=True, use_graph=True)
    with torch.no_grad():
        exp = render_rays(models, emb, rays, ts, S, False, 0, 0, I, chunk, True, True)
    for k in exp:
        assert torch.equal(third[k], exp[k]), k


def test_camera_rays_under_a_graph():
    """CameraRays + use_graph: the captured prologue reads the camera from device memory (C ABI nfl_pass_args::d_cam), so
    one capture renders any pose / pixel range -- bit-identical to the eager camera prologue (VERDICT r2 weak #10d)."""
    import gpu_util
    from nerf_fl_amd import CameraRays
    from nerf_fl_amd.eval import batched_inference, fov60_intrinsics
    from nerf_fl_amd.poses import make_c2w
    dev = gpu_util.DEV
    models, emb = _nerfw_models(dev)
    H, W, S, I, chunk = 21, 30, 32, 32, 256
    K = fov60_intrinsics(W, H)
    cache = {}
    for k, eye in enumerate(([0.1, -0.2, 3.9], [0.6, 0.3, 3.5])):
        c2w = make_c2w(torch.tensor([0.1, -0.2, 0.05]), torch.tensor(eye))[:3]
        cam = CameraRays(c2w, K, H, W, 2.0, 6.0, dev)
        ts = torch.full((H * W,), 5 + k, dtype=torch.long, device=dev)
        got = batched_inference(models, emb, cam, ts, S, I, chunk=chunk, white_back=True, use_graph=True, _graph_cache=cache)
        exp = batched_inference(models, emb, cam, ts, S, I, chunk=chunk, white_back=True, use_graph=False)
        assert len(cache) == 1
        for key in exp:
            assert torch.equal(got[key], exp[key]), key
