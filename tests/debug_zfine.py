"""Debug helper: where do the HIP fine depths differ from the reference's, and what does the reference's cdf look like there?"""
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
got = gpu_util.hip_render(specs, a["rays"], kw, precision="f16x3", field_raw=True)
zf, zr = got["_z_fine"], a["z_fine"]
d = (zf - zr).abs()
print("max |z_fine diff|", d.max().item(), "rays with diff > 1e-5:", (d.max(1)[0] > 1e-5).nonzero().flatten().tolist())
cond = gu.fixture_conditioning(cfg, a)
rays = a["rays"]; rnd = gu.random_inputs(cfg, a)
z = orc.coarse_depths(rays[:, 6:7], rays[:, 7:8], cfg["S"], cfg["use_disp"], cfg["perturb"], rnd["perturb_rand"])
wref, whip = a["out.weights_coarse"], got["weights_coarse"]
I = cfg["I"]
for r in (d.max(1)[0] > 1e-5).nonzero().flatten().tolist()[:4]:
    print(f"--- ray {r}: conditioning {cond[r].item():.2e}; coarse weight max diff {(wref[r]-whip[r]).abs().max().item():.2e}")
    for w, tag in ((wref[r:r+1], "ref"), (whip[r:r+1], "hip")):
        mids = 0.5 * (z[r:r+1, :-1] + z[r:r+1, 1:])
        u = rnd["u"][r:r+1] if cfg["perturb"] > 0 else torch.linspace(0, 1, I)[None]
        s = orc.sample_pdf(mids, w[:, 1:-1], u)
        ww = w[:, 1:-1] + 1e-5
        tot = ww.sum(1)
        pdf = ww / tot[:, None]
        cdf = torch.cat([torch.zeros(1, 1), pdf.cumsum(1)], 1)
        print(f"   [{tag}] total {tot.item():.9f}  cdf[-3:] {cdf[0, -3:].tolist()}  samples via CPU sample_pdf of these weights: first diff vs fixture order stats:")
        zz = torch.sort(torch.cat([z[r:r+1], s], 1), 1)[0]
        print(f"        max diff to fixture z_fine {(zz - zr[r:r+1]).abs().max().item():.3e}; to HIP z_fine {(zz - zf[r:r+1]).abs().max().item():.3e}")
    idx = d[r].argmax().item()
    print("   z_fine hip/ref around the worst index", idx, zf[r, max(idx-2,0):idx+3].tolist(), zr[r, max(idx-2,0):idx+3].tolist())
    print("   ref interior weights (x1e6):", [round(v * 1e6, 2) for v in wref[r, 1:-1].tolist()][:70])
print("=== all rays: cond, max z diff, max transient_sigmas diff")
ts_d = (got["transient_sigmas"] - a["out.transient_sigmas"]).abs().max(1)[0] if "transient_sigmas" in got else torch.zeros(zf.shape[0])
for r in range(zf.shape[0]):
    if d[r].max() > 1e-6 or ts_d[r] > 1e-4 or cond[r] <= 1e-4:
        print(f"ray {r:3d} cond {cond[r].item():.2e} zdiff {d[r].max().item():.2e} at {d[r].argmax().item()} tsig diff {ts_d[r].item():.2e}")
