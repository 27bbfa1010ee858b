"""Debug helper (not a test): per-sample field outputs of the HIP kernel vs the oracle MLP on a fixture, with the
oracle's intermediate magnitudes of the worst samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import golden_util as gu
import gpu_util
from oracle import nerfw_oracle as orc

name = sys.argv[1] if len(sys.argv) > 1 else "g17_trained_cfg3"
cfg, a = gu.load(name)
specs, kw = gu.oracle_kwargs(cfg, a)
spec_c, P_c, spec_f, P_f = specs
got = gpu_util.hip_render(specs, a["rays"], kw, precision="f16x3", field_raw=True)
rays = a["rays"]
z = got["_z_fine"]
R, F = z.shape
xyz = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]
enc = orc.posenc(xyz.reshape(-1, 3), spec_f.n_emb_xyz)
side = [orc.posenc(rays[:, 3:6], 4)]
if spec_f.encode_appearance:
    side.append(kw["a_emb"])
dir_a = torch.cat(side, 1).repeat_interleave(F, 0)
tau = kw["t_emb"].repeat_interleave(F, 0)
P = P_f
with torch.no_grad():
    # manual forward keeping intermediates
    h = enc
    acts = {}
    for i in range(8):
        if i == 4:
            h = torch.cat([enc, h], 1)
        h = torch.relu(orc._lin(P, f"xyz_encoding_{i+1}.0", h))
        acts[f"h{i+1}"] = h
    feat = orc._lin(P, "xyz_encoding_final", h)
    acts["feat"] = feat
    g = torch.cat([feat, tau], 1)
    for j in (0, 2, 4, 6):
        pre = orc._lin(P, f"transient_encoding.{j}", g)
        acts[f"tpre{j}"] = pre
        g = torch.relu(pre)
        acts[f"g{j}"] = g
    pre_s = orc._lin(P, "transient_sigma.0", g)[:, 0]
    o = orc.field_forward(spec_f, P, enc, dir_a, tau)
raw = got["_field_raw_fine"]
for k, col in (("sigma", 3), ("sigma_t", 7), ("beta", 8)):
    err = (raw[:, col] - o[k]).abs()
    print(k, "max err", err.max().item(), "at", err.argmax().item(), "n > 1e-4:", int((err > 1e-4).sum()), "of", err.numel())
for k, sl in (("rgb", slice(0, 3)), ("rgb_t", slice(4, 7))):
    err = (raw[:, sl] - o[k]).abs().max(1)[0]
    print(k, "max err", err.max().item(), "n > 1e-4:", int((err > 1e-4).sum()))
err = (raw[:, 7] - o["sigma_t"]).abs()
worst = err.argsort(descending=True)[:8]
for w in worst.tolist():
    print(f"sample {w} (ray {w // F}, i {w % F}): sigma_t hip {raw[w,7].item():.6f} ref {o['sigma_t'][w].item():.6f} pre_s {pre_s[w].item():.4f} "
          + " ".join(f"{k}:max|.|={v[w].abs().max().item():.2f}" for k, v in acts.items()))
print("global max |act|:", {k: round(v.abs().max().item(), 2) for k, v in acts.items()})
# distribution of errors vs magnitude of feat
big = err > 1e-4
print("bad samples: count", int(big.sum()), " feat max over bad:", acts["feat"][big].abs().max().item() if big.any() else None)
print("tau used (first bad ray):", kw["t_emb"][worst[0] // F])
