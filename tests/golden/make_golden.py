#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference); the GPU box never
sees the reference, only the small .npz files this script writes.  The fixtures
hold inputs and expected outputs only -- weights are re-created on both sides
from `oracle.nerfw_oracle.make_field_params(spec, seed, regime)` (numpy PCG64,
platform independent) and pushed into the reference modules with
load_state_dict(), so no reference source or checkpoint is stored here.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [prefix ...]   (e.g. g15_ g17_: only those cases)
        PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py train            (w_trained.npz: the reference's own
                                                                                        weights after 400 Adam steps;
                                                                                        run it before the g17_* cases)
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from models.nerf import NeRF, PosEmbedding, BarfPosEmbedding   # noqa: E402  (reference)
from models.rendering import render_rays, sample_pdf  # noqa: E402  (reference)
from losses import loss_dict                           # noqa: E402  (reference)

from oracle import nerfw_oracle as orc                 # noqa: E402

torch.set_num_threads(8)


def ref_field(spec: orc.FieldSpec, seed: int, regime: str, refine_pose: bool = False):
    m = NeRF(spec.typ, in_channels_xyz=spec.c_xyz, in_channels_dir=spec.c_dir,
             encode_appearance=spec.encode_appearance, in_channels_a=spec.n_a,
             encode_transient=spec.encode_transient, in_channels_t=spec.n_tau,
             beta_min=spec.beta_min, refine_pose=refine_pose)
    P = trained_params(spec) if regime == "trained" else orc.make_field_params(spec, seed, regime)
    missing = m.load_state_dict(P, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


TRAINED = os.path.join(HERE, "w_trained.npz")


def trained_params(spec: orc.FieldSpec):
    """Weights of the reference after `train_reference()` (tests/golden/w_trained.npz): the coarse field for every
    base-shaped field, the NeRF-W fine field otherwise."""
    z = np.load(TRAINED, allow_pickle=False)
    tag = "fine" if (spec.encode_appearance or spec.encode_transient) else "coarse"
    if tag == "fine":
        assert spec.encode_appearance and spec.encode_transient
    return {k[len(tag) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + ".")}


def save(name, cfg, **arrays):
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
           for k, v in arrays.items()}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), cfg=json.dumps(cfg), **out)
    print(f"wrote {name}.npz  ({len(out)} arrays)")


class CaptureRandom:
    """Record every torch.rand_like / randn_like / rand the reference draws."""

    def __enter__(self):
        self.log = []
        self._orig = (torch.rand_like, torch.randn_like, torch.rand)

        def wrap(fn, tag):
            def inner(*a, **k):
                out = fn(*a, **k)
                self.log.append((tag, out.clone()))
                return out
            return inner

        torch.rand_like = wrap(self._orig[0], "rand_like")
        torch.randn_like = wrap(self._orig[1], "randn_like")
        torch.rand = wrap(self._orig[2], "rand")
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.randn_like, torch.rand = self._orig


# ----------------------------------------------------------------------------
def g1_posenc():
    rng = np.random.default_rng(101)
    x = torch.from_numpy((rng.uniform(-6, 6, size=(64, 3))).astype(np.float32))
    arrays = {"x": x}
    for n in (10, 4, 15):
        arrays[f"out_{n}"] = PosEmbedding(n - 1, n)(x)
    save("g1_posenc", {"n_freqs": [10, 4, 15]}, **arrays)


def g2_field():
    rng = np.random.default_rng(202)
    B = 256
    for tag, spec, st, ot in [
        ("sigma", orc.FieldSpec("coarse"), True, False),
        ("base", orc.FieldSpec("coarse"), False, False),
        ("a", orc.FieldSpec("fine", encode_appearance=True), False, False),
        ("at", orc.FieldSpec("fine", encode_appearance=True, encode_transient=True), False, True),
    ]:
        for regime in ("default", "sharp"):
            m = ref_field(spec, 7, regime)
            xyz = torch.from_numpy(rng.uniform(-3, 3, size=(B, 3)).astype(np.float32))
            cols = [PosEmbedding(spec.n_emb_xyz - 1, spec.n_emb_xyz)(xyz)]
            if not st:
                d = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((B, 3)).astype(np.float32)), dim=1)
                cols.append(PosEmbedding(spec.n_emb_dir - 1, spec.n_emb_dir)(d))
                if spec.encode_appearance:
                    cols.append(torch.from_numpy(rng.standard_normal((B, spec.n_a)).astype(np.float32)))
                if ot:
                    cols.append(torch.from_numpy(rng.standard_normal((B, spec.n_tau)).astype(np.float32)))
            x = torch.cat(cols, 1)
            with torch.no_grad():
                y = m(x, sigma_only=True) if st else m(x, output_transient=ot)
            save(f"g2_field_{tag}_{regime}",
                 {"spec": spec.__dict__, "seed": 7, "regime": regime, "sigma_only": st, "output_transient": ot},
                 x=x, y=y)


def g3_sample_pdf():
    rng = np.random.default_rng(303)
    R, M, I = 48, 62, 64
    bins = np.sort(rng.uniform(2, 6, size=(R, M + 1)).astype(np.float32), axis=1)
    w = rng.uniform(0, 1, size=(R, M)).astype(np.float32) ** 4
    w[:8, 10:40] = 0.0              # runs of empty bins
    w[8:12, :] = 0.0                # completely empty rays
    w[12:16, :] = 0.0
    w[12:16, 31] = 1.0              # one spike
    bins_t, w_t = torch.from_numpy(bins), torch.from_numpy(w)
    det = sample_pdf(bins_t, w_t, I, det=True)
    u = rng.uniform(0, 1, size=(R, I)).astype(np.float32)
    u[:, 0] = 0.0
    u[:, 1] = 1.0
    u[:, 2] = np.float32(1.0) - np.float32(2 ** -24)
    u_t = torch.from_numpy(u)
    orig = torch.rand
    torch.rand = lambda *a, **k: u_t.clone()
    try:
        rnd = sample_pdf(bins_t, w_t, I, det=False)
    finally:
        torch.rand = orig
    save("g3_sample_pdf", {"n_importance": I}, bins=bins_t, weights=w_t, u=u_t, det=det, rnd=rnd)


def g3b_sample_pdf_from_coarse():
    """sample_pdf exactly as render_rays calls it (rendering.py:266-272): bins = mid-points of the coarse depths,
    weights = the interior coarse weights, then concat + sort.  Same edge cases as g3 (empty rays, runs of zero
    weight, a spike, u = 0 / 1 / 1 - 2^-24) plus per-ray depth ranges; this is what the C ABI's nfl_sample_pdf
    takes, so the HIP kernel can be tested on it directly."""
    rng = np.random.default_rng(404)
    R, S, I = 48, 64, 64
    near = rng.uniform(0.05, 2.0, size=(R, 1)).astype(np.float32)
    far = (near + rng.uniform(1.0, 4.0, size=(R, 1))).astype(np.float32)
    steps = np.linspace(0, 1, S, dtype=np.float32)[None]
    z = near * (1 - steps) + far * steps
    mid = 0.5 * (z[:, :-1] + z[:, 1:])
    lower, upper = np.concatenate([z[:, :1], mid], 1), np.concatenate([mid, z[:, -1:]], 1)
    z[24:] = (lower + (upper - lower) * rng.uniform(0, 1, size=(R, S)).astype(np.float32))[24:]   # jittered rows
    w = rng.uniform(0, 1, size=(R, S)).astype(np.float32) ** 4
    w[:8, 10:40] = 0.0              # runs of empty bins
    w[8:12, :] = 0.0                # completely empty rays
    w[12:16, :] = 0.0
    w[12:16, 31] = 1.0              # one spike
    w[16:20, 1:-1] = 0.0            # weight only in the two columns sample_pdf drops
    z_t, w_t = torch.from_numpy(z), torch.from_numpy(w)
    mids = 0.5 * (z_t[:, :-1] + z_t[:, 1:])
    det = sample_pdf(mids, w_t[:, 1:-1], I, det=True)
    u = rng.uniform(0, 1, size=(R, I)).astype(np.float32)
    u[:, 0] = 0.0
    u[:, 1] = 1.0
    u[:, 2] = np.float32(1.0) - np.float32(2 ** -24)
    u_t = torch.from_numpy(u)
    orig = torch.rand
    torch.rand = lambda *a, **k: u_t.clone()
    try:
        rnd = sample_pdf(mids, w_t[:, 1:-1], I, det=False)
    finally:
        torch.rand = orig
    save("g3b_sample_pdf_coarse", {"n_samples": S, "n_importance": I}, z_coarse=z_t, weights_coarse=w_t, u=u_t,
         det=det, rnd=rnd, z_fine_det=torch.sort(torch.cat([z_t, det], -1), -1)[0],
         z_fine_rnd=torch.sort(torch.cat([z_t, rnd], -1), -1)[0])


def _scene_colors(rays, ts, n_vocab, rng):
    """Analytic target for train_reference(): a shaded unit sphere in front of a white background, a per-image tint
    (appearance) and, on every third image, a grey occluder over part of the view (transient)."""
    o, d = rays[:, :3].numpy().astype(np.float64), rays[:, 3:6].numpy().astype(np.float64)
    b = (o * d).sum(1)
    disc = b * b - ((o * o).sum(1) - 1.0)
    hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0.0))
    n = o + d * t[:, None]
    stripes = 0.5 + 0.5 * np.sign(np.sin(9.0 * n[:, 0]) * np.sin(9.0 * n[:, 1]))
    col = np.clip(0.5 + 0.5 * n, 0, 1) * (0.55 + 0.45 * stripes[:, None])
    tint = 0.75 + 0.25 * np.random.default_rng(5).uniform(-1, 1, size=(n_vocab, 3))
    col = np.where(hit[:, None], col * tint[ts.numpy()], 1.0)
    occ = (ts.numpy() % 3 == 0) & (d[:, 0] > 0.05)
    col = np.where(occ[:, None], 0.35, col)
    return torch.from_numpy(np.clip(col, 0, 1).astype(np.float32))


def train_reference(steps=400, R=512, n_vocab=20, lr=1e-3):
    """'Trained-like' weights (random-init weights have a narrower dynamic range than trained ones): fit
    the REAL reference (coarse base field + NeRF-W fine field + latent tables, NerfWLoss, Adam) to an analytic scene
    for a few hundred steps on the CPU and store the resulting state_dicts.  The low 8 mantissa bits of every weight
    are cleared before storing (keeps the file compressible; the fixtures are generated FROM the stored values, so
    nothing is lost)."""
    torch.manual_seed(77)
    spec_c = orc.FieldSpec("coarse")
    spec_f = orc.FieldSpec("fine", encode_appearance=True, encode_transient=True, beta_min=0.1)
    mc, mf = ref_field(spec_c, 71, "default"), ref_field(spec_f, 72, "default")
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4), "a": torch.nn.Embedding(n_vocab, 48),
           "t": torch.nn.Embedding(n_vocab, 16)}
    params = list(mc.parameters()) + list(mf.parameters()) + list(emb["a"].parameters()) + list(emb["t"].parameters())
    opt = torch.optim.Adam(params, lr=lr, eps=1e-8)
    loss_fn = loss_dict["nerfw"]()
    rng = np.random.default_rng(78)
    for it in range(steps):
        rays = orc.make_rays(R, 1000 + it)
        ts = torch.from_numpy(rng.integers(0, n_vocab, size=R).astype(np.int64))
        target = _scene_colors(rays, ts, n_vocab, rng)
        opt.zero_grad()
        res = render_rays({"coarse": mc, "fine": mf}, emb, rays, ts, 64, False, 1.0, 1.0, 64, 32768, True, False)
        losses = loss_fn(res, target)
        sum(losses.values()).backward()
        opt.step()
        if it % 20 == 0 or it == steps - 1:
            mse = ((res["rgb_fine"].detach() - target) ** 2).mean().item()
            print(f"train_reference step {it}: " + " ".join(f"{k}={float(v):.4f}" for k, v in losses.items())
                  + f" psnr_fine={-10 * np.log10(mse):.2f}", flush=True)
    out = {}
    for tag, m in (("coarse", mc), ("fine", mf)):
        for k, v in m.state_dict().items():
            bits = v.detach().numpy().view(np.uint32) & np.uint32(0xFFFFFF00)
            out[f"{tag}.{k}"] = bits.view(np.float32)
    for k in ("a", "t"):
        out[f"table_{k}"] = emb[k].weight.detach().numpy()
    np.savez_compressed(TRAINED, **out)
    print("wrote", TRAINED)


def render_case(name, *, R, S, I, fine=None, regime="sharp", white_back=True, use_disp=False,
                perturb=0.0, noise_std=0.0, test_time=False, n_vocab=20, kwargs_mode="ts",
                output_transient=None, grads=False, rays_grad=False, near=2.0, far=6.0, seed=11,
                n_emb_xyz=10, barf_epoch=None, rays_kind="blender", view_dir=False, beta_min=0.1, n_emb_dir=4,
                n_a=48, n_tau=16):
    """fine: None | 'base' | 'a' | 'at'.  rays_kind 'photo': per-ray near/far (phototourism); view_dir: pass a
    `view_dir` kwarg that differs from rays_d (rendering.py:236-238)."""
    spec_c = orc.FieldSpec("coarse", n_emb_xyz=n_emb_xyz, n_emb_dir=n_emb_dir)
    barf = barf_epoch is not None
    mc = ref_field(spec_c, seed, regime, refine_pose=barf)
    models = {"coarse": mc}
    if barf:        # train.py:42-44
        embeddings = {"xyz": BarfPosEmbedding(n_emb_xyz - 1, n_emb_xyz, 4, 8),
                      "dir": BarfPosEmbedding(n_emb_dir - 1, n_emb_dir, 4, 8)}
    else:
        embeddings = {"xyz": PosEmbedding(n_emb_xyz - 1, n_emb_xyz), "dir": PosEmbedding(n_emb_dir - 1, n_emb_dir)}
    cfg = dict(R=R, S=S, I=I, fine=fine, regime=regime, white_back=white_back, use_disp=use_disp,
               perturb=perturb, noise_std=noise_std, test_time=test_time, n_vocab=n_vocab,
               kwargs_mode=kwargs_mode, output_transient=output_transient, seed=seed,
               n_emb_xyz=n_emb_xyz, n_emb_dir=n_emb_dir, beta_min=beta_min, barf_epoch=barf_epoch, rays_kind=rays_kind,
               n_a=n_a, n_tau=n_tau)
    spec_f = None
    if fine is not None:
        spec_f = orc.FieldSpec("fine", n_emb_xyz=n_emb_xyz, n_emb_dir=n_emb_dir, encode_appearance=fine in ("a", "at"),
                               encode_transient=fine == "at", beta_min=beta_min, n_a=n_a, n_tau=n_tau)
        models["fine"] = ref_field(spec_f, seed + 1, regime, refine_pose=barf)
    rays = orc.make_rays_photo(R, seed + 2) if rays_kind == "photo" else orc.make_rays(R, seed + 2, near, far)
    rng = np.random.default_rng(seed + 3)
    ts = torch.from_numpy(rng.integers(0, n_vocab, size=R).astype(np.int64))
    arrays = {"rays": rays, "ts": ts}
    kwargs = {}
    if barf:
        kwargs["current_epoch"] = barf_epoch
    a_emb = t_emb = None
    if spec_f is not None and spec_f.encode_appearance:
        table_a = (torch.from_numpy(np.load(TRAINED)["table_a"]) if regime == "trained"
                   else orc.make_embedding_table(n_vocab, n_a, seed + 4))
        emb = torch.nn.Embedding(n_vocab, n_a)
        emb.weight.data.copy_(table_a)
        embeddings["a"] = emb
        a_emb = table_a[ts].clone()
    if spec_f is not None and spec_f.encode_transient:
        table_t = (torch.from_numpy(np.load(TRAINED)["table_t"]) if regime == "trained"
                   else orc.make_embedding_table(n_vocab, n_tau, seed + 5))
        emb = torch.nn.Embedding(n_vocab, n_tau)
        emb.weight.data.copy_(table_t)
        embeddings["t"] = emb
        t_emb = table_t[ts].clone()
    if kwargs_mode == "embedded":       # the notebooks' a_embedded / t_embedded kwargs
        if a_emb is not None:
            a_emb.requires_grad_(grads)
            kwargs["a_embedded"] = a_emb
        if t_emb is not None:
            t_emb.requires_grad_(grads)
            kwargs["t_embedded"] = t_emb
    if output_transient is not None:
        kwargs["output_transient"] = output_transient
    if view_dir:
        vd = np.random.default_rng(seed + 7).standard_normal((R, 3))
        vd /= np.linalg.norm(vd, axis=1, keepdims=True)
        arrays["view_dir"] = kwargs["view_dir"] = torch.from_numpy(vd.astype(np.float32))
    if rays_grad:
        rays.requires_grad_(True)
    for p in list(mc.parameters()) + (list(models["fine"].parameters()) if fine else []):
        p.requires_grad_(grads)

    torch.manual_seed(1000 + seed + R + S + I)     # the captured draws are part of the fixture: keep them reproducible
    sorted_rows, orig_sort = [], torch.sort

    def sort_spy(*a, **k):         # the reference sorts once: the merged fine depths (rendering.py:272)
        out = orig_sort(*a, **k)
        sorted_rows.append(out[0].detach().clone())
        return out

    torch.sort = sort_spy
    try:
        with CaptureRandom() as cap:
            ctx = torch.enable_grad() if grads else torch.no_grad()
            with ctx:
                res = render_rays(models, embeddings, rays, ts, S, use_disp, perturb, noise_std, I,
                                  32768, white_back, test_time, **kwargs)
    finally:
        torch.sort = orig_sort
    if I > 0:
        assert len(sorted_rows) == 1 and tuple(sorted_rows[0].shape) == (R, S + I)
        arrays["z_fine"] = sorted_rows[0]          # the depths the reference's fine pass used
    cfg["rng_order"] = [t for t, _ in cap.log]
    cfg["keys"] = list(res.keys())
    for i, (tag, val) in enumerate(cap.log):
        arrays[f"rng{i}_{tag}"] = val
    for k, v in res.items():
        arrays["out." + k] = v

    if grads:
        rng_t = np.random.default_rng(seed + 6)
        target = torch.from_numpy(rng_t.uniform(0, 1, size=(R, 3)).astype(np.float32))
        arrays["target"] = target
        loss = sum(loss_dict["nerfw"]()(res, target).values())
        loss.backward()
        arrays["loss"] = loss.detach()
        for tag, m in models.items():
            for n, p in m.named_parameters():
                g = p.grad
                if g is None:
                    continue
                if g.numel() <= 4096:
                    arrays[f"grad.{tag}.{n}"] = g
                else:
                    arrays[f"gradrows.{tag}.{n}"] = g[:4]
                    arrays[f"gradnorm.{tag}.{n}"] = g.norm()
                    pr = torch.from_numpy(np.random.default_rng(99).standard_normal(g.numel()).astype(np.float32))
                    arrays[f"gradproj.{tag}.{n}"] = (g.flatten() * pr).sum()
                    # a second, independent projection and the first four COLUMNS (input features; the rows above are output
                    # features): VERDICT r2 weak #5
                    pr2 = torch.from_numpy(np.random.default_rng(100).standard_normal(g.numel()).astype(np.float32))
                    arrays[f"gradproj2.{tag}.{n}"] = (g.flatten() * pr2).sum()
                    arrays[f"gradcols.{tag}.{n}"] = g[:, :4].contiguous()
        if kwargs_mode == "embedded":
            if a_emb is not None:
                arrays["grad.a_emb"] = a_emb.grad
            if t_emb is not None:
                arrays["grad.t_emb"] = t_emb.grad
        else:
            for k in ("a", "t"):
                if k in embeddings and embeddings[k].weight.grad is not None:
                    arrays[f"grad.table_{k}"] = embeddings[k].weight.grad
        if rays_grad:
            arrays["grad.rays"] = rays.grad
    save(name, cfg, **arrays)


def main():
    want = sys.argv[1:]
    sel = lambda name: not want or any(name.startswith(w) for w in want)
    if "train" in want:
        train_reference()
        return
    if sel("g1_"):
        g1_posenc()
    if sel("g2_"):
        g2_field()
    if sel("g3_"):
        g3_sample_pdf()
    if sel("g3b_"):
        g3b_sample_pdf_from_coarse()
    for name, kw in RENDER_CASES:
        if sel(name):
            render_case(name, **kw)


RENDER_CASES = [
    # G4: cfg-1 shape, coarse only
    ("g4_cfg1_coarse", dict(R=64, S=32, I=0, white_back=True)),
    ("g4_cfg1_coarse_default", dict(R=64, S=32, I=0, white_back=True, regime="default")),
    # G5: cfg-2 shape, base coarse + base fine
    ("g5_cfg2_base", dict(R=64, S=64, I=64, fine="base", white_back=True)),
    ("g5_cfg2_base_default", dict(R=64, S=64, I=64, fine="base", white_back=True, regime="default")),
    # G6: cfg-3 shape, NeRF-W a+t, train keys
    ("g6_cfg3_nerfw", dict(R=64, S=64, I=64, fine="at", white_back=True)),
    ("g6_cfg3_nerfa", dict(R=64, S=64, I=64, fine="a", white_back=True)),
    # G7: test_time keys; output_transient=False + a_embedded kwarg
    ("g7_test_nerfw", dict(R=64, S=64, I=64, fine="at", white_back=True, test_time=True)),
    ("g7_test_nerfw_noT", dict(R=64, S=64, I=64, fine="at", white_back=False, test_time=True,
                               kwargs_mode="embedded", output_transient=False)),
    # G8: disparity sampling; G9: black background, near/far 0.5/5 (the same on every ray; per-ray bounds: G15)
    ("g8_use_disp", dict(R=64, S=64, I=64, fine="base", white_back=True, use_disp=True)),
    ("g9_black_back", dict(R=64, S=64, I=64, fine="at", white_back=False, near=0.5, far=5.0)),
    # G10: cfg-5 shape 128+128, sigma-only coarse, N_emb_xyz=15 like the phototourism notebook
    ("g10_cfg5", dict(R=32, S=128, I=128, fine="at", white_back=False, test_time=True, near=0.3, far=5.0)),
    ("g10_cfg5_xyz15", dict(R=32, S=128, I=128, fine="a", white_back=False, test_time=True,
                            n_emb_xyz=15, near=0.3, far=5.0)),
    # odd sizes: ragged sample counts (reference default N_importance=128 with 64 coarse)
    ("g13_ragged", dict(R=37, S=24, I=40, fine="base", white_back=True)),
    ("g13_64_128", dict(R=33, S=64, I=128, fine="at", white_back=False)),
    # G11: gradients
    ("g11_grad_cfg1", dict(R=64, S=32, I=0, white_back=True, grads=True)),
    ("g11_grad_cfg2", dict(R=64, S=64, I=64, fine="base", white_back=True, grads=True)),
    ("g11_grad_cfg3", dict(R=64, S=64, I=64, fine="at", white_back=True, grads=True, kwargs_mode="embedded")),
    ("g11_grad_cfg3_ts", dict(R=64, S=64, I=64, fine="at", white_back=False, grads=True)),
    ("g11_grad_rays", dict(R=32, S=32, I=32, fine="base", white_back=True, grads=True, rays_grad=True)),
    # G12: stochastic runs with captured draws
    ("g12_stoch_base", dict(R=64, S=64, I=64, fine="base", white_back=True, perturb=1.0, noise_std=1.0)),
    ("g12_stoch_nerfw", dict(R=64, S=64, I=64, fine="at", white_back=True, perturb=1.0, noise_std=1.0)),
    ("g12_stoch_grad", dict(R=64, S=64, I=64, fine="base", white_back=True, perturb=1.0, noise_std=1.0, grads=True)),
    # G14: learnable-pose mode: BARF-weighted encodings (epochs inside and after the ramp) + gradient w.r.t. rays
    ("g14_barf_e6", dict(R=32, S=32, I=32, fine="base", white_back=True, grads=True, rays_grad=True, barf_epoch=6)),
    ("g14_barf_e9", dict(R=32, S=64, I=64, fine="at", white_back=False, grads=True, rays_grad=True, barf_epoch=9,
                         kwargs_mode="embedded")),
    ("g14_barf_e2_fwd", dict(R=32, S=32, I=32, fine="base", white_back=True, barf_epoch=2)),
    # G15: configs[3] on one GPU -- Phototourism shape (README.md:113-120): batch 1024, 64+64, NeRF-W a+t, black
    # background, N_vocab 1500 through the embedding tables, beta_min 0.03, near/far different on every ray;
    # deterministic and stochastic, forward + gradients (incl. the dense table gradients)
    ("g15_photo_grad", dict(R=1024, S=64, I=64, fine="at", white_back=False, n_vocab=1500, beta_min=0.03,
                            rays_kind="photo", grads=True, noise_std=1.0, seed=15)),
    ("g15_photo_stoch", dict(R=256, S=64, I=64, fine="at", white_back=False, n_vocab=1500, beta_min=0.03,
                             rays_kind="photo", perturb=1.0, noise_std=1.0, grads=True, seed=16)),
    ("g15_photo_test", dict(R=128, S=128, I=128, fine="at", white_back=False, n_vocab=1500, beta_min=0.03,
                            rays_kind="photo", test_time=True, seed=17, use_disp=True)),
    # G16: the view_dir kwarg (rendering.py:236-238): direction encoding of a vector other than rays_d
    ("g16_view_dir", dict(R=64, S=64, I=64, fine="a", white_back=True, view_dir=True, grads=True, seed=18)),
    ("g16_view_dir_test", dict(R=48, S=32, I=32, fine="at", white_back=False, view_dir=True, test_time=True, seed=19,
                               rays_kind="photo")),
    # G17: weights after a few hundred Adam steps of the reference (train_reference): cfg 2 and cfg 3, forward
    # (deterministic, stochastic, test_time) and gradients
    # G18: other encoder widths (opt.py:25-28 takes any integer): xyz 6 / dir 2, xyz 12 / dir 4, xyz 3 / dir 1
    ("g18_emb6_2", dict(R=48, S=32, I=32, fine="at", white_back=True, grads=True, n_emb_xyz=6, n_emb_dir=2, seed=31)),
    ("g18_emb12_4", dict(R=48, S=32, I=32, fine="base", white_back=False, grads=True, rays_grad=True, n_emb_xyz=12, seed=32)),
    ("g18_emb6_2_test", dict(R=40, S=64, I=64, fine="at", white_back=False, test_time=True, n_emb_xyz=6, n_emb_dir=2, seed=33,
                             near=0.3, far=5.0)),
    ("g18_emb3_1_barf", dict(R=32, S=32, I=32, fine="base", white_back=True, grads=True, rays_grad=True, barf_epoch=6,
                             n_emb_xyz=3, n_emb_dir=1, seed=34)),
    ("g18_emb14_3_stoch", dict(R=48, S=64, I=64, fine="a", white_back=True, perturb=1.0, noise_std=1.0, n_emb_xyz=14,
                               n_emb_dir=3, seed=35)),
    # other latent widths (opt.py --N_a / --N_tau): 24 / 8 through the tables, 40 / 5 through the a_embedded / t_embedded kwargs
    ("g18_na24_tau8", dict(R=48, S=32, I=32, fine="at", white_back=True, grads=True, n_a=24, n_tau=8, seed=37)),
    ("g18_na40_tau5_emb", dict(R=48, S=32, I=32, fine="at", white_back=False, grads=True, kwargs_mode="embedded", n_a=40, n_tau=5,
                               n_emb_xyz=12, seed=38)),
    # view_dir given AND a gradient w.r.t. the rays (rendering.py:236-238: the direction encoding then does not depend on rays)
    ("g16_view_dir_rays", dict(R=48, S=32, I=32, fine="a", white_back=True, view_dir=True, grads=True, rays_grad=True, seed=36)),
    ("g17_trained_cfg2", dict(R=64, S=64, I=64, fine="base", white_back=True, regime="trained", grads=True, seed=21)),
    ("g17_trained_cfg3", dict(R=64, S=64, I=64, fine="at", white_back=True, regime="trained", grads=True, seed=22)),
    ("g17_trained_cfg3_stoch", dict(R=64, S=64, I=64, fine="at", white_back=True, regime="trained", perturb=1.0,
                                    noise_std=1.0, seed=23)),
    ("g17_trained_cfg2_stoch", dict(R=64, S=64, I=64, fine="base", white_back=True, regime="trained", perturb=1.0,
                                    noise_std=1.0, grads=True, seed=24)),
    ("g17_trained_test", dict(R=32, S=128, I=128, fine="at", white_back=True, regime="trained", test_time=True,
                              seed=25)),
]


if __name__ == "__main__":
    main()
