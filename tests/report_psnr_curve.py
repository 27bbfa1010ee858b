"""Report (not a test; run on the GPU box): per-step training loss of the HIP run of tests/test_psnr_parity_gpu.py (base) against the
stored reference curve and its replicas, in windows of 10 steps over the first 200 steps."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import golden_util as gu
import gpu_util
import psnr_scene as sc
from oracle import nerfw_oracle as orc
from nerf_fl_amd import PosEmbedding, render_rays
from nerf_fl_amd.train import Adam, NerfWLoss

kind = "base"
use_torch_adam = len(sys.argv) > 1 and sys.argv[1] == "torchadam"
dev = gpu_util.DEV
ref = np.load(os.path.join(gu.GOLDEN_DIR, f"psnr_{kind}.npz"))
reps = [np.load(os.path.join(gu.GOLDEN_DIR, f"psnr_{kind}_replica{s}.npz")) for s in ("", "2")]
cfg = sc.CONFIGS[kind]
S, I, R, steps = cfg["S"], cfg["I"], cfg["R"], 300
spec_c, spec_f = orc.FieldSpec("coarse"), orc.FieldSpec("fine", beta_min=0.1)
models = {"coarse": gpu_util.module_from(spec_c, orc.make_field_params(spec_c, cfg["seed"], "default")),
          "fine": gpu_util.module_from(spec_f, orc.make_field_params(spec_f, cfg["seed"] + 1, "default"))}
emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
params = [p for m in models.values() for p in m.parameters()]
opt = torch.optim.Adam(params, lr=cfg["lr"], eps=1e-8) if use_torch_adam else Adam(params, lr=cfg["lr"], eps=1e-8)
loss_fn = NerfWLoss()
losses = []
for it in range(steps):
    rays, ts, target = sc.batch(cfg, it)
    d = {k: v.to(dev) for k, v in sc.draws(cfg, it).items()}
    for grp in opt.param_groups:
        grp["lr"] = sc.cosine_lr(cfg, it)
    opt.zero_grad(set_to_none=True)
    res = render_rays(models, emb, rays.to(dev), ts.to(dev), S, False, 1.0, 1.0, I, 32768, True, False, **d)
    loss = sum(loss_fn(res, target.to(dev)).values())
    loss.backward()
    opt.step()
    if use_torch_adam:
        torch.autograd.graph.increment_version(params)
    losses.append(loss.detach())
L = torch.stack(losses).cpu().numpy()
w = 10
wm = lambda x: np.asarray(x)[:steps].reshape(-1, w).mean(1)
a, b, r1, r2 = wm(L), wm(ref["losses"]), wm(reps[0]["losses"]), wm(reps[1]["losses"])
print("adam:", "torch.optim.Adam" if use_torch_adam else "nerf_fl_amd Adam")
print("window  hip/ref-1 [%]   replica/ref-1 [%]   replica2/ref-1 [%]")
for i in range(len(a)):
    print(f"{i * w:4d}   {100 * (a[i] / b[i] - 1):8.2f}   {100 * (r1[i] / b[i] - 1):8.2f}   {100 * (r2[i] / b[i] - 1):8.2f}")
print("per-step |loss_hip - loss_ref| / loss_ref over the first 12 steps:", np.round(np.abs(L[:12] - ref["losses"][:12]) / ref["losses"][:12], 6))
