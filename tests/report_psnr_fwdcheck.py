"""Along a HIP training trajectory (64+64 base fit), every 50 steps: the loss of the SAME weights / batch / draws computed by
the oracle's fp32 forward on the GPU, and the cosine between the HIP gradient and the oracle-autograd gradient, overall and
per tensor (not a test).   python tests/report_psnr_fwdcheck.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_util
import psnr_scene as sc
from oracle import nerfw_oracle as orc
from nerf_fl_amd import PosEmbedding, render_rays
from nerf_fl_amd.train import Adam

cfg = sc.CONFIGS["base"]
S, I, steps = cfg["S"], cfg["I"], cfg["steps"]
dev = gpu_util.DEV
spec_c, spec_f = orc.FieldSpec("coarse"), orc.FieldSpec("fine")
models = {"coarse": gpu_util.module_from(spec_c, orc.make_field_params(spec_c, cfg["seed"], "default")),
          "fine": gpu_util.module_from(spec_f, orc.make_field_params(spec_f, cfg["seed"] + 1, "default"))}
emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
named = [(f"{t}.{n}", p) for t, m in models.items() for n, p in m.named_parameters()]
params = [p for _, p in named]
opt = Adam(params, lr=cfg["lr"], eps=1e-8)
for it in range(steps):
    rays, ts, target = sc.batch(cfg, it)
    d = {k: v.to(dev) for k, v in sc.draws(cfg, it).items()}
    for grp in opt.param_groups:
        grp["lr"] = sc.cosine_lr(cfg, it)
    opt.zero_grad(set_to_none=True)
    res = render_rays(models, emb, rays.to(dev), ts.to(dev), S, False, 1.0, 1.0, I, 32768, True, False, **d)
    loss = sum(orc.nerfw_loss(res, target.to(dev)).values())
    loss.backward()
    if it % 50 == 0 or it % 50 == 1 or it == steps - 1:
        # the oracle on the same weights (detached copies with their own autograd graph)
        Pc = {n: p.detach().clone().requires_grad_(True) for n, p in models["coarse"].named_parameters()}
        Pf = {n: p.detach().clone().requires_grad_(True) for n, p in models["fine"].named_parameters()}
        torch.set_default_device(dev)
        ro = orc.render_rays(spec_c, Pc, spec_f, Pf, rays.to(dev), n_samples=S, n_importance=I, perturb=1.0, noise_std=1.0,
                             white_back=True, **d)
        lo = sum(orc.nerfw_loss(ro, target.to(dev)).values())
        lo.backward()
        torch.set_default_device("cpu")
        go = {f"coarse.{n}": p.grad for n, p in Pc.items()}
        go.update({f"fine.{n}": p.grad for n, p in Pf.items()})
        gh = {n: p.grad for n, p in named}
        worst = []
        tot_dot = tot_a = tot_b = 0.0
        for n in gh:
            a, b = gh[n].double().flatten(), go[n].double().flatten()
            dot, na, nb = float(a @ b), float(a.norm()), float(b.norm())
            tot_dot, tot_a, tot_b = tot_dot + dot, tot_a + na * na, tot_b + nb * nb
            worst.append((1 - dot / (na * nb + 1e-300), float((a - b).norm() / (nb + 1e-300)), n))
        worst.sort(reverse=True)
        # coherence of the error from one step to the next: cosine between e_t = g_hip - g_autograd at steps 50k and 50k+1
        err = {n: (gh[n].double() - go[n].double()).flatten() for n in gh}
        if it % 50 == 1:
            coh = []
            for n in err:
                a, b = err[n], prev_err[n]
                coh.append((float(a @ b / (a.norm() * b.norm() + 1e-300)), n))
            coh.sort(reverse=True)
            allc = torch.cat([err[n] for n in err]) @ torch.cat([prev_err[n] for n in err]) / (
                torch.cat([err[n] for n in err]).norm() * torch.cat([prev_err[n] for n in err]).norm())
            # a scale error shows as a component of e along g
            along = float(torch.cat([err[n] for n in err]) @ torch.cat([go[n].double().flatten() for n in err])
                          / torch.cat([go[n].double().flatten() for n in err]).norm() ** 2)
            print(f"   error coherence steps {it - 1}->{it}: all tensors {float(allc):+.3f}; along-gradient component {along:+.2e}; most coherent: "
                  + "; ".join(f"{n} {c:+.2f}" for c, n in coh[:4]) + " | least: " + "; ".join(f"{n} {c:+.2f}" for c, n in coh[-2:]), flush=True)
        prev_err = err
        print(f"step {it:3d}: loss hip {float(loss):.6f} oracle {float(lo):.6f} rel {abs(float(loss) - float(lo)) / float(lo):.1e}; "
              f"grad cos deficit {1 - tot_dot / (tot_a * tot_b) ** 0.5:.2e}; worst tensors (1-cos, relL2): "
              + "; ".join(f"{n} {c:.1e} {e:.1e}" for c, e, n in worst[:3]), flush=True)
    opt.step()
