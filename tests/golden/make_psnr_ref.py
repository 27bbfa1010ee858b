#!/usr/bin/env python3
"""Train the REAL reference (models/nerf.py + models/rendering.py + losses.py, CPU) on a small analytic scene and store
its loss curve and validation PSNR: the anchor of tests/test_psnr_parity_gpu.py, which repeats the identical run -- same
initial weights, same ray batches, same random draws, same Adam -- with the HIP renderer and compares.

Only data is written (tests/golden/psnr_*.npz): loss per step, validation PSNR, the hyper-parameters and seeds.  Every
random draw comes from a numpy PCG64 stream keyed by (seed, step, kind), so the GPU side regenerates the same numbers.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_psnr_ref.py [base|nerfw]
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from models.nerf import NeRF, PosEmbedding            # noqa: E402  (reference)
from models.rendering import render_rays              # noqa: E402  (reference)
from losses import loss_dict                          # noqa: E402  (reference)

import psnr_scene as sc                               # noqa: E402  (tests/psnr_scene.py: inputs shared with the GPU test)
from oracle import nerfw_oracle as orc                # noqa: E402

torch.set_num_threads(int(os.environ.get("NFL_REF_THREADS", "8")))


class InjectRandom:
    """Serve torch.rand_like / randn_like / rand inside the reference from a prepared list, in call order."""

    def __init__(self, draws):
        self.draws = list(draws)

    def __enter__(self):
        self._orig = (torch.rand_like, torch.randn_like, torch.rand)
        pop = lambda *a, **k: self.draws.pop(0)
        torch.rand_like = torch.randn_like = torch.rand = pop
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.randn_like, torch.rand = self._orig
        assert not self.draws, "the reference drew fewer tensors than prepared"


def run(kind, perturb=0.0, replica=1, grad_noise=0.0):
    """perturb > 0: a replica whose initial weights are multiplied by (1 + perturb * N(0,1)) -- how far the REFERENCE
    moves from itself under rounding-level differences; stored as psnr_<kind>_replica[2].npz."""
    cfg = sc.CONFIGS[kind]
    S, I, R, steps = cfg["S"], cfg["I"], cfg["R"], cfg["steps"]
    nerfw = cfg["fine"] == "at"
    spec_c = orc.FieldSpec("coarse")
    spec_f = orc.FieldSpec("fine", encode_appearance=nerfw, encode_transient=nerfw, beta_min=0.1)
    mc = NeRF("coarse")
    mc.load_state_dict(orc.make_field_params(spec_c, cfg["seed"], "default"))
    mf = NeRF("fine", encode_appearance=nerfw, in_channels_a=48, encode_transient=nerfw, in_channels_t=16, beta_min=0.1)
    mf.load_state_dict(orc.make_field_params(spec_f, cfg["seed"] + 1, "default"))
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    params = list(mc.parameters()) + list(mf.parameters())
    if perturb > 0:
        g = torch.Generator().manual_seed(4242 + 1000 * (replica - 1))
        with torch.no_grad():
            for p_ in params:
                p_.mul_(1 + perturb * torch.randn(p_.shape, generator=g))
    if nerfw:
        for k, dim, off in (("a", 48, 4), ("t", 16, 5)):
            e = torch.nn.Embedding(cfg["n_vocab"], dim)
            e.weight.data.copy_(orc.make_embedding_table(cfg["n_vocab"], dim, cfg["seed"] + off))
            emb[k] = e
            params += list(e.parameters())
    opt = torch.optim.Adam(params, lr=cfg["lr"], eps=1e-8)
    gn = torch.Generator().manual_seed(777 + replica)
    loss_fn = loss_dict["nerfw"]()
    models = {"coarse": mc, "fine": mf}
    losses = []
    for it in range(steps):
        rays, ts, target = sc.batch(cfg, it)
        d = sc.draws(cfg, it)
        seq = [d["perturb_rand"], d["noise_coarse"], d["u"]] + ([] if nerfw else [d["noise_fine"]])
        for grp in opt.param_groups:
            grp["lr"] = sc.cosine_lr(cfg, it)
        opt.zero_grad()
        with InjectRandom(seq):
            res = render_rays(models, emb, rays, ts, S, False, 1.0, 1.0, I, 32768, True, False)
        loss = sum(loss_fn(res, target).values())
        loss.backward()
        if grad_noise > 0:      # experiment: a noise floor of grad_noise * max|g| per tensor on every gradient, as a
            for p_ in params:   # reduced-precision backward leaves (tests/test_psnr_parity_gpu.py discusses the outcome)
                if p_.grad is not None:
                    p_.grad.add_(grad_noise * p_.grad.abs().max() * torch.randn(p_.grad.shape, generator=gn))
        opt.step()
        losses.append(float(loss))
        if it % 25 == 0 or it == steps - 1:
            print(f"[{kind}] step {it}: loss {losses[-1]:.5f}", flush=True)
    rays, ts, target = sc.val_batch(cfg)
    with torch.no_grad(), InjectRandom([torch.zeros(rays.shape[0], S)] + ([] if nerfw else [torch.zeros(rays.shape[0], S + I)])):
        res = render_rays(models, emb, rays, ts, S, False, 0, 0.0, I, 32768, True, False)
    psnr = float(-10.0 * torch.log10(((res["rgb_fine"] - target) ** 2).mean()))
    print(f"[{kind}] validation PSNR of the reference-trained model: {psnr:.3f} dB")
    tag = f"psnr_{kind}" + (("_replica" + ("" if replica == 1 else str(replica))) if perturb > 0 else "")
    if grad_noise > 0:
        tag = f"psnr_{kind}_gradnoise{replica}"
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), cfg=json.dumps(cfg), losses=np.asarray(losses, np.float32),
                        val_psnr=np.float32(psnr), perturb=np.float32(perturb), grad_noise=np.float32(grad_noise))


if __name__ == "__main__":
    import re
    for kind in (sys.argv[1:] or ["base", "nerfw"]):
        m = re.fullmatch(r"(base|nerfw|smooth)_(replica|gradnoise)(\d*)", kind)
        if m and m.group(2) == "replica":          # base_replica, base_replica2, base_replica3, ...
            run(m.group(1), perturb=1e-6, replica=int(m.group(3) or 1))
        elif m:                                    # base_gradnoise1, ...: the reference with a 1e-3 gradient noise floor
            run(m.group(1), replica=int(m.group(3) or 1), grad_noise=1e-3)
        else:
            run(kind)
