#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference); the GPU box never
sees the reference, only the small .npz files this script writes.  The fixtures
hold inputs and expected outputs only -- weights are re-created on both sides
from `oracle.nerfw_oracle.make_field_params(spec, seed, regime)` (numpy PCG64,
platform independent) and pushed into the reference modules with
load_state_dict(), so no reference source or checkpoint is stored here.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
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
    P = orc.make_field_params(spec, seed, regime)
    missing = m.load_state_dict(P, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


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


def render_case(name, *, R, S, I, fine=None, regime="sharp", white_back=True, use_disp=False,
                perturb=0.0, noise_std=0.0, test_time=False, n_vocab=20, kwargs_mode="ts",
                output_transient=None, grads=False, rays_grad=False, near=2.0, far=6.0, seed=11,
                n_emb_xyz=10, barf_epoch=None):
    """fine: None | 'base' | 'a' | 'at'."""
    spec_c = orc.FieldSpec("coarse", n_emb_xyz=n_emb_xyz)
    barf = barf_epoch is not None
    mc = ref_field(spec_c, seed, regime, refine_pose=barf)
    models = {"coarse": mc}
    if barf:        # train.py:42-44
        embeddings = {"xyz": BarfPosEmbedding(n_emb_xyz - 1, n_emb_xyz, 4, 8), "dir": BarfPosEmbedding(3, 4, 4, 8)}
    else:
        embeddings = {"xyz": PosEmbedding(n_emb_xyz - 1, n_emb_xyz), "dir": PosEmbedding(3, 4)}
    cfg = dict(R=R, S=S, I=I, fine=fine, regime=regime, white_back=white_back, use_disp=use_disp,
               perturb=perturb, noise_std=noise_std, test_time=test_time, n_vocab=n_vocab,
               kwargs_mode=kwargs_mode, output_transient=output_transient, seed=seed,
               n_emb_xyz=n_emb_xyz, beta_min=0.1, barf_epoch=barf_epoch)
    spec_f = None
    if fine is not None:
        spec_f = orc.FieldSpec("fine", n_emb_xyz=n_emb_xyz, encode_appearance=fine in ("a", "at"),
                               encode_transient=fine == "at", beta_min=0.1)
        models["fine"] = ref_field(spec_f, seed + 1, regime, refine_pose=barf)
    rays = orc.make_rays(R, seed + 2, near, far)
    rng = np.random.default_rng(seed + 3)
    ts = torch.from_numpy(rng.integers(0, n_vocab, size=R).astype(np.int64))
    arrays = {"rays": rays, "ts": ts}
    kwargs = {}
    if barf:
        kwargs["current_epoch"] = barf_epoch
    a_emb = t_emb = None
    if spec_f is not None and spec_f.encode_appearance:
        table_a = orc.make_embedding_table(n_vocab, 48, seed + 4)
        emb = torch.nn.Embedding(n_vocab, 48)
        emb.weight.data.copy_(table_a)
        embeddings["a"] = emb
        a_emb = table_a[ts].clone()
    if spec_f is not None and spec_f.encode_transient:
        table_t = orc.make_embedding_table(n_vocab, 16, seed + 5)
        emb = torch.nn.Embedding(n_vocab, 16)
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
    if rays_grad:
        rays.requires_grad_(True)
    for p in list(mc.parameters()) + (list(models["fine"].parameters()) if fine else []):
        p.requires_grad_(grads)

    torch.manual_seed(1000 + seed + R + S + I)     # the captured draws are part of the fixture: keep them reproducible
    with CaptureRandom() as cap:
        ctx = torch.enable_grad() if grads else torch.no_grad()
        with ctx:
            res = render_rays(models, embeddings, rays, ts, S, use_disp, perturb, noise_std, I,
                              32768, white_back, test_time, **kwargs)
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
    g1_posenc()
    g2_field()
    g3_sample_pdf()
    # G4: cfg-1 shape, coarse only
    render_case("g4_cfg1_coarse", R=64, S=32, I=0, white_back=True)
    render_case("g4_cfg1_coarse_default", R=64, S=32, I=0, white_back=True, regime="default")
    # G5: cfg-2 shape, base coarse + base fine
    render_case("g5_cfg2_base", R=64, S=64, I=64, fine="base", white_back=True)
    render_case("g5_cfg2_base_default", R=64, S=64, I=64, fine="base", white_back=True, regime="default")
    # G6: cfg-3 shape, NeRF-W a+t, train keys
    render_case("g6_cfg3_nerfw", R=64, S=64, I=64, fine="at", white_back=True)
    render_case("g6_cfg3_nerfa", R=64, S=64, I=64, fine="a", white_back=True)
    # G7: test_time keys; output_transient=False + a_embedded kwarg
    render_case("g7_test_nerfw", R=64, S=64, I=64, fine="at", white_back=True, test_time=True)
    render_case("g7_test_nerfw_noT", R=64, S=64, I=64, fine="at", white_back=False, test_time=True,
                kwargs_mode="embedded", output_transient=False)
    # G8: disparity sampling; G9: black background, per-ray near/far like phototourism
    render_case("g8_use_disp", R=64, S=64, I=64, fine="base", white_back=True, use_disp=True)
    render_case("g9_black_back", R=64, S=64, I=64, fine="at", white_back=False, near=0.5, far=5.0)
    # G10: cfg-5 shape 128+128, sigma-only coarse, N_emb_xyz=15 like the phototourism notebook
    render_case("g10_cfg5", R=32, S=128, I=128, fine="at", white_back=False, test_time=True, near=0.3, far=5.0)
    render_case("g10_cfg5_xyz15", R=32, S=128, I=128, fine="a", white_back=False, test_time=True,
                n_emb_xyz=15, near=0.3, far=5.0)
    # odd sizes: ragged sample counts (reference default N_importance=128 with 64 coarse)
    render_case("g13_ragged", R=37, S=24, I=40, fine="base", white_back=True)
    render_case("g13_64_128", R=33, S=64, I=128, fine="at", white_back=False)
    # G11: gradients
    render_case("g11_grad_cfg1", R=64, S=32, I=0, white_back=True, grads=True)
    render_case("g11_grad_cfg2", R=64, S=64, I=64, fine="base", white_back=True, grads=True)
    render_case("g11_grad_cfg3", R=64, S=64, I=64, fine="at", white_back=True, grads=True,
                kwargs_mode="embedded")
    render_case("g11_grad_cfg3_ts", R=64, S=64, I=64, fine="at", white_back=False, grads=True)
    render_case("g11_grad_rays", R=32, S=32, I=32, fine="base", white_back=True, grads=True, rays_grad=True)
    # G12: stochastic runs with captured draws
    render_case("g12_stoch_base", R=64, S=64, I=64, fine="base", white_back=True, perturb=1.0, noise_std=1.0)
    render_case("g12_stoch_nerfw", R=64, S=64, I=64, fine="at", white_back=True, perturb=1.0, noise_std=1.0)
    render_case("g12_stoch_grad", R=64, S=64, I=64, fine="base", white_back=True, perturb=1.0, noise_std=1.0,
                grads=True)
    # G14: learnable-pose mode: BARF-weighted encodings (epochs inside and after the ramp) + gradient w.r.t. rays
    render_case("g14_barf_e6", R=32, S=32, I=32, fine="base", white_back=True, grads=True, rays_grad=True, barf_epoch=6)
    render_case("g14_barf_e9", R=32, S=64, I=64, fine="at", white_back=False, grads=True, rays_grad=True, barf_epoch=9,
                kwargs_mode="embedded")
    render_case("g14_barf_e2_fwd", R=32, S=32, I=32, fine="base", white_back=True, barf_epoch=2)


if __name__ == "__main__":
    main()
