"""Gradient parity of the HIP backward against the CPU oracle + autograd for configurations the
golden gradient fixtures do not cover (ragged sample counts, rays spanning several tiles,
NeRF-A, transient head switched off on a NeRF-W model, black background)."""
import pytest
import torch

from oracle import nerfw_oracle as orc

pytestmark = pytest.mark.gpu
GTOL = 1e-2

CASES = {
    "ragged_24_40": dict(S=24, I=40, fine="base", white=True, perturb=1.0, noise_std=1.0, out_t=True),
    "spans_tiles_64_128": dict(S=64, I=128, fine="at", white=False, perturb=1.0, noise_std=0.0, out_t=True),
    "nerf_a": dict(S=32, I=32, fine="a", white=True, perturb=0.0, noise_std=1.0, out_t=True),
    "nerfw_transient_off": dict(S=32, I=32, fine="at", white=False, perturb=0.0, noise_std=1.0, out_t=False),
    "coarse_only_96": dict(S=96, I=0, fine=None, white=True, perturb=1.0, noise_std=1.0, out_t=True),
}


@pytest.mark.parametrize("backward", ["f16", "f16w", "f16x3"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_gradients_vs_oracle_autograd(name, backward):
    """Every backward arithmetic on the shapes the fixtures do not have: ragged sample counts (padded segments write the
    scratch record of the -- possibly split -- gradient stash), rays spanning several tiles, heads switched off."""
    import nerf_fl_amd
    nerf_fl_amd.set_precision("f16x3", backward=backward)
    try:
        _gradients_vs_oracle_autograd(name)
    finally:
        nerf_fl_amd.set_precision("f16x3", backward="f16")


def _gradients_vs_oracle_autograd(name):
    import gpu_util
    from nerf_fl_amd import PosEmbedding, render_rays
    c = CASES[name]
    dev = gpu_util.DEV
    R, S, I = 40, c["S"], c["I"]
    F = S + I
    g = torch.Generator().manual_seed(123)
    spec_c = orc.FieldSpec("coarse")
    P_c = orc.make_field_params(spec_c, 51, "sharp")
    spec_f = P_f = None
    if c["fine"]:
        spec_f = orc.FieldSpec("fine", encode_appearance=c["fine"] in ("a", "at"), encode_transient=c["fine"] == "at",
                               beta_min=0.1)
        P_f = orc.make_field_params(spec_f, 52, "sharp")
    rays = orc.make_rays(R, 53)
    target = torch.rand(R, 3, generator=g)
    a_emb = torch.randn(R, 48, generator=g) if spec_f is not None and spec_f.encode_appearance else None
    t_emb = torch.randn(R, 16, generator=g) if spec_f is not None and spec_f.encode_transient else None
    rnd = dict(perturb_rand=torch.rand(R, S, generator=g) if c["perturb"] > 0 else None,
               noise_coarse=torch.randn(R, S, generator=g),
               u=torch.rand(R, I, generator=g) if (I > 0 and c["perturb"] > 0) else None,
               noise_fine=torch.randn(R, F, generator=g) if I > 0 else None)
    use_t = bool(t_emb is not None and c["out_t"])
    if use_t:
        rnd["noise_fine"] = None

    # ---- oracle + autograd
    leaves = {}
    for tag, P in (("coarse", P_c), ("fine", P_f)):
        if P is not None:
            for n, p in P.items():
                leaves[f"{tag}.{n}"] = p.requires_grad_(True)
    a_o = a_emb.clone().requires_grad_(True) if a_emb is not None else None
    t_o = t_emb.clone().requires_grad_(True) if t_emb is not None else None
    res = orc.render_rays(spec_c, P_c, spec_f, P_f, rays, n_samples=S, n_importance=I, perturb=c["perturb"],
                          noise_std=c["noise_std"], white_back=c["white"], a_emb=a_o, t_emb=t_o,
                          output_transient=c["out_t"], **rnd)
    loss_o = sum(orc.nerfw_loss(res, target).values())
    loss_o.backward()

    # ---- HIP
    models = {"coarse": gpu_util.module_from(spec_c, {k: v.detach() for k, v in P_c.items()})}
    if spec_f is not None:
        models["fine"] = gpu_util.module_from(spec_f, {k: v.detach() for k, v in P_f.items()})
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    extra = {k: v.to(dev) for k, v in rnd.items() if v is not None}
    a_h = a_emb.to(dev).requires_grad_(True) if a_emb is not None else None
    t_h = t_emb.to(dev).requires_grad_(True) if t_emb is not None else None
    if a_h is not None:
        extra["a_embedded"] = a_h
    if t_h is not None:
        extra["t_embedded"] = t_h
    if not c["out_t"]:
        extra["output_transient"] = False
    out = render_rays(models, emb, rays.to(dev), torch.zeros(R, dtype=torch.long, device=dev), S, False, c["perturb"],
                      c["noise_std"], I, 32768, c["white"], False, **extra)
    assert list(out.keys()) == list(res.keys())
    loss_h = sum(orc.nerfw_loss(out, target.to(dev)).values())
    loss_h.backward()
    assert abs(float(loss_h.detach()) - float(loss_o.detach())) <= 1e-4 * max(1.0, abs(float(loss_o.detach())))

    bad = {}
    def check(key, got, exp):
        if exp is None:
            assert got is None or float(got.abs().max()) == 0.0, key
            return
        ref = exp.abs().max().item()
        err = (got.cpu() - exp).abs().max().item()
        if not err <= GTOL * ref + 1e-7:
            bad[key] = (err, ref)
    for tag, m in models.items():
        for n, p in m.named_parameters():
            check(f"{tag}.{n}", p.grad, leaves[f"{tag}.{n}"].grad)
    if a_h is not None:
        check("a_emb", a_h.grad, a_o.grad)
    if t_h is not None:
        if use_t:
            check("t_emb", t_h.grad, t_o.grad)
        else:
            assert t_h.grad is None
    assert not bad, bad


def test_learnable_pose_gradients_end_to_end():
    """--refine_pose: (r, t) -> c2w -> rays -> BARF-encoded render -> loss; the pose gradients that come
    back through the HIP backward's d/d rays match the CPU oracle + autograd."""
    import gpu_util
    from nerf_fl_amd import render_rays
    from nerf_fl_amd.poses import LearnPose, get_ray_directions, get_rays
    dev = gpu_util.DEV
    H = W = 6
    Kmat = torch.tensor([[7.0, 0, 3.0], [0, 7.0, 3.0], [0, 0, 1]])
    dirs = get_ray_directions(H, W, Kmat).reshape(-1, 3)                       # 36 rays of one camera
    init = torch.eye(4)[None].repeat(2, 1, 1)
    init[:, :3, 3] = torch.tensor([[0.1, -0.1, 4.0], [0.3, 0.2, 4.2]])
    cam = torch.tensor([0] * 18 + [1] * 18)
    spec_c, spec_f = orc.FieldSpec("coarse"), orc.FieldSpec("fine")
    P_c, P_f = orc.make_field_params(spec_c, 71, "sharp"), orc.make_field_params(spec_f, 72, "sharp")
    target = torch.rand(36, 3, generator=torch.Generator().manual_seed(3))
    epoch, S, I = 6, 32, 32

    def rays_from(pose, device):
        c2w = pose(cam.to(device))
        o, d = get_rays(dirs.to(device), c2w)
        nf = torch.tensor([2.0, 6.0], device=device).expand(36, 2)
        return torch.cat([o, d, nf], 1)

    # oracle
    pose_o = LearnPose(2, True, True, init_c2w=init)
    with torch.no_grad():
        pose_o.r.add_(torch.tensor([[0.02, -0.01, 0.03], [-0.02, 0.01, 0.0]]))
        pose_o.t.add_(torch.tensor([[0.01, 0.0, -0.02], [0.0, 0.02, 0.01]]))
    res = orc.render_rays(spec_c, P_c, spec_f, P_f, rays_from(pose_o, "cpu"), n_samples=S, n_importance=I,
                          noise_std=0.0, white_back=True, pe_w_xyz=orc.barf_weights(10, epoch),
                          pe_w_dir=orc.barf_weights(4, epoch))
    sum(orc.nerfw_loss(res, target).values()).backward()

    # HIP
    pose_h = LearnPose(2, True, True, init_c2w=init).to(dev)
    pose_h.load_state_dict(pose_o.state_dict())
    models = {"coarse": gpu_util.module_from(spec_c, P_c, True), "fine": gpu_util.module_from(spec_f, P_f, True)}
    for m in models.values():
        m.requires_grad_(False)                                                # only the poses are optimised here
    emb = gpu_util.make_embeddings(10, True)
    out = render_rays(models, emb, rays_from(pose_h, dev), torch.zeros(36, dtype=torch.long, device=dev), S, False, 0,
                      0.0, I, 32768, True, False, current_epoch=epoch)
    sum(orc.nerfw_loss(out, target.to(dev)).values()).backward()
    for name in ("r", "t"):
        g_h, g_o = getattr(pose_h, name).grad.cpu(), getattr(pose_o, name).grad
        assert g_o.abs().max() > 0
        assert (g_h - g_o).abs().max().item() <= GTOL * g_o.abs().max().item(), (name, g_h, g_o)


@pytest.mark.parametrize("log2_factor", [-30, 20])
def test_loss_scale_makes_backward_magnitude_invariant(log2_factor):
    """The MLP backward runs in fp16 under a device-chosen power-of-two loss scale.  A loss
    multiplied by 2^k must therefore give gradients multiplied by 2^k -- far outside fp16's own range -- up to the
    summation-order noise of the fp32 atomics."""
    import gpu_util
    from nerf_fl_amd import PosEmbedding, render_rays
    dev = gpu_util.DEV
    R, S, I = 64, 32, 32
    spec_c, spec_f = orc.FieldSpec("coarse"), orc.FieldSpec("fine", encode_appearance=True, encode_transient=True, beta_min=0.1)
    P_c, P_f = orc.make_field_params(spec_c, 61, "sharp"), orc.make_field_params(spec_f, 62, "sharp")
    g = torch.Generator().manual_seed(7)
    rays, target = orc.make_rays(R, 63).to(dev), torch.rand(R, 3, generator=g).to(dev)
    a_emb, t_emb = torch.randn(R, 48, generator=g).to(dev), torch.randn(R, 16, generator=g).to(dev)
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    ts = torch.zeros(R, dtype=torch.long, device=dev)

    def grads(factor):
        models = {"coarse": gpu_util.module_from(spec_c, P_c), "fine": gpu_util.module_from(spec_f, P_f)}
        a_h, t_h = a_emb.clone().requires_grad_(True), t_emb.clone().requires_grad_(True)
        out = render_rays(models, emb, rays, ts, S, False, 0.0, 0.0, I, 32768, True, False, a_embedded=a_h, t_embedded=t_h)
        (sum(orc.nerfw_loss(out, target).values()) * factor).backward()
        gs = {f"{t}.{n}": p.grad for t, m in models.items() for n, p in m.named_parameters() if p.grad is not None}
        gs["a"], gs["t"] = a_h.grad, t_h.grad
        return gs

    base, scaled = grads(1.0), grads(2.0 ** log2_factor)
    assert set(base) == set(scaled)
    for k in base:
        assert torch.isfinite(scaled[k]).all(), k
        ref = base[k].abs().max().item()
        err = (scaled[k] * 2.0 ** (-log2_factor) - base[k]).abs().max().item()
        assert err <= 1e-4 * ref + 1e-12, (k, err, ref)
