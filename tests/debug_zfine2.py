import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import golden_util as gu
import gpu_util
from oracle import nerfw_oracle as orc
torch.set_printoptions(precision=9, linewidth=200)
name, r = sys.argv[1], int(sys.argv[2])
cfg, a = gu.load(name)
specs, kw = gu.oracle_kwargs(cfg, a)
got = gpu_util.hip_render(specs, a["rays"], kw, precision="f16x3", field_raw=True)
zf, zr = got["_z_fine"][r], a["z_fine"][r]
rays = a["rays"]; rnd = gu.random_inputs(cfg, a)
z = orc.coarse_depths(rays[:, 6:7], rays[:, 7:8], cfg["S"], cfg["use_disp"], cfg["perturb"], rnd["perturb_rand"])[r:r+1]
I = cfg["I"]
u = rnd["u"][r:r+1] if cfg["perturb"] > 0 else torch.linspace(0, 1, I)[None]
mids = 0.5 * (z[:, :-1] + z[:, 1:])
for tag, w in (("ref", a["out.weights_coarse"][r:r+1]), ("hip", got["weights_coarse"][r:r+1])):
    ww = w[:, 1:-1] + 1e-5
    tot = ww.sum(1); pdf = ww / tot[:, None]
    cdf = torch.cat([torch.zeros(1, 1), pdf.cumsum(1)], 1)
    s = orc.sample_pdf(mids, w[:, 1:-1], u)
    hi = torch.searchsorted(cdf, u.contiguous(), right=True)
    print(tag, "total", tot.item())
    if tag == "ref":
        s_ref, cdf_ref, hi_ref = s, cdf, hi
    else:
        dd = (s - s_ref).abs()[0]
        for i in dd.argsort(descending=True)[:5].tolist():
            j = hi_ref[0, i].item() - 1
            print(f"  draw {i} u {u[0,i].item():.9f}: ref sample {s_ref[0,i].item():.7f} hip-weights sample {s[0,i].item():.7f} diff {dd[i].item():.2e}; ref bin {j} (hip bin {hi[0,i].item()-1}) "
                  f"cdf_ref[j..j+1] {cdf_ref[0,j].item():.9f} {cdf_ref[0,min(j+1,62)].item():.9f} den {cdf_ref[0,min(j+1,62)].item()-cdf_ref[0,j].item():.3e} "
                  f"cdf_hip[j..j+1] {cdf[0,j].item():.9f} {cdf[0,min(j+1,62)].item():.9f} w_ref[j] {a['out.weights_coarse'][r, j+1].item():.3e} w_hip[j] {got['weights_coarse'][r, j+1].item():.3e}")
print("z_fine hip vs ref max diff", (zf - zr).abs().max().item())
