"""The 64+64 PSNR-parity fit with the ORACLE (plain fp32 torch ops + autograd, oracle/nerfw_oracle.py) running on the GPU
instead of the HIP renderer (not a test): separates "GPU arithmetic / harness" from "hand-written kernels" when the HIP
fit's loss curve drifts from the CPU reference's.   python tests/report_psnr_oracle_gpu.py [n] [adam=torch|hip]"""
import glob
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import golden_util as gu
import psnr_scene as sc
from oracle import nerfw_oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
which_adam = ([a[5:] for a in sys.argv[2:] if a.startswith("adam=")] or ["torch"])[0]
rel_noise = float(([a[4:] for a in sys.argv[2:] if a.startswith("rel=")] or ["0"])[0])
frozen_noise = float(([a[7:] for a in sys.argv[2:] if a.startswith("frozen=")] or ["0"])[0])
elem_noise = float(([a[5:] for a in sys.argv[2:] if a.startswith("elem=")] or ["0"])[0])
# bwd=w16: every Linear's backward multiplies by the fp16-ROUNDED weight (what the HIP dgrad does; the rounding pattern is
# persistent); bwd=a16: the weight gradient uses fp16-rounded input activations and output gradients (what the HIP wgrad does)
bwd_mode = ([a[4:] for a in sys.argv[2:] if a.startswith("bwd=")] or [""])[0]


_scale = [None]


class _LinearLowBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        wq = w.half().float() if "w16" in bwd_mode else w
        if "chain" in bwd_mode:
            # one loss scale per step, fixed by the first (head) layer the backward reaches, like the HIP pass' d_gmax;
            # the fp16-rounded gradient is what propagates AND what enters the weight gradient
            if _scale[0] is None:
                _scale[0] = 32.0 / g.abs().max().clamp_min(1e-30)
            s = _scale[0]
            gq, xq = (g * s).half().float() / s, x.half().float()
            gx = gq @ wq
        elif "a16" in bwd_mode:
            s = 32.0 / g.abs().max().clamp_min(1e-30)                  # the loss scale: max -> 2^5
            gq, xq = (g * s).half().float() / s, x.half().float()
            gx = g @ wq
        else:
            gq, xq = g, x
            gx = g @ wq
        g2, x2 = gq.reshape(-1, gq.shape[-1]), xq.reshape(-1, xq.shape[-1])
        return gx, g2.t() @ x2, g2.sum(0)


if bwd_mode:
    orc._lin = lambda P, name, x: _LinearLowBwd.apply(x, P[name + ".weight"], P[name + ".bias"])
kind = "base"
cfg = sc.CONFIGS[kind]
S, I, steps = cfg["S"], cfg["I"], cfg["steps"]
win = 50
wmean = lambda x: np.asarray(x, np.float64)[: len(x) // win * win].reshape(-1, win).mean(1)
refs = [np.load(f, allow_pickle=False) for f in sorted(glob.glob(os.path.join(gu.GOLDEN_DIR, f"psnr_{kind}*.npz"))) if "gradnoise" not in f]
Rm = np.stack([wmean(r["losses"]) for r in refs]).mean(0)
dev = torch.device("cuda", 0)
spec_c, spec_f = orc.FieldSpec("coarse"), orc.FieldSpec("fine")
for run in range(n):
    P_c = {k: v.to(dev).requires_grad_(True) for k, v in orc.make_field_params(spec_c, cfg["seed"], "default").items()}
    P_f = {k: v.to(dev).requires_grad_(True) for k, v in orc.make_field_params(spec_f, cfg["seed"] + 1, "default").items()}
    params = list(P_c.values()) + list(P_f.values())
    if which_adam == "hip":
        from nerf_fl_amd.train import Adam
        opt = Adam(params, lr=cfg["lr"], eps=1e-8)
    else:
        opt = torch.optim.Adam(params, lr=cfg["lr"], eps=1e-8)
    losses = []
    torch.set_default_device(dev)
    for it in range(steps):
        rays, ts, target = sc.batch(cfg, it)
        d = {k: v.to(dev) for k, v in sc.draws(cfg, it).items()}
        for grp in opt.param_groups:
            grp["lr"] = sc.cosine_lr(cfg, it)
        opt.zero_grad(set_to_none=True)
        res = orc.render_rays(spec_c, P_c, spec_f, P_f, rays.to(dev), n_samples=S, n_importance=I, perturb=1.0, noise_std=1.0,
                              white_back=True, **d)
        loss = sum(orc.nerfw_loss(res, target.to(dev)).values())
        _scale[0] = None
        loss.backward()
        if rel_noise > 0:      # white noise of relative l2 size rel_noise on every gradient tensor, fresh every step
            for p_ in params:
                p_.grad.add_(rel_noise * p_.grad.norm() / p_.grad.numel() ** 0.5 * torch.randn_like(p_.grad))
        if frozen_noise > 0:   # the same noise pattern at every step (a persistent error, like rounded weights in a backward)
            if it == 0:
                gz = torch.Generator(device=dev).manual_seed(1234 + run)
                Z = [torch.randn(p_.shape, generator=gz, device=dev) for p_ in params]
            for p_, z_ in zip(params, Z):
                p_.grad.add_(frozen_noise * p_.grad.norm() / p_.grad.numel() ** 0.5 * z_)
        if elem_noise > 0:     # every element multiplied by (1 + elem_noise * N(0,1))
            for p_ in params:
                p_.grad.mul_(1 + elem_noise * torch.randn_like(p_.grad))
        opt.step()
        losses.append(loss.detach())
    torch.set_default_device("cpu")
    losses = torch.stack(losses).cpu().numpy()
    print(f"oracle-on-GPU ({which_adam} Adam, bwd {bwd_mode or chr(45)}, rel {rel_noise} frozen {frozen_noise} elem {elem_noise}) run {run}: first loss {losses[0]:.6f} (ref {refs[0]['losses'][0]:.6f}); dev (%)",
          " ".join(f"{100 * (a - b) / b:7.2f}" for a, b in zip(wmean(losses), Rm)), flush=True)
