"""Is the stochastic rounding of the f16 / f16w backward unbiased?  (run on the GPU box; not a test)
python tests/report_sr_mean.py [case] [n_seeds]
One fixture, its gradients computed with n different rounding seeds: the l2 error (vs the reference's fp32 autograd gradient
the fixture holds) of every single draw, and of the MEAN over the draws.  Zero-mean independent rounding errors shrink like
1/sqrt(n) in the mean; what does not shrink is rounding that is not redrawn with the seed (the fp16 activation stash, and in
'f16' the transposed weights, whose draw is a function of the weight bits) plus the reference's own fp32 noise."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_fl_amd
import test_grad_gpu as T

name = sys.argv[1] if len(sys.argv) > 1 else "g11_grad_cfg3_ts"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for backward in ("f16", "f16w"):
    draws = []
    for seed in range(n):
        nerf_fl_amd.set_rounding_seed(seed)
        cfg, a, got, loss = T.run_case(name, backward)
        draws.append(got)
    nerf_fl_amd.set_rounding_seed(0)
    keys = [k[5:] for k in a if k.startswith("grad.") and k != "grad.rays" and a[k].abs().max() > 0 and a[k].numel() > 64]
    print(f"== {name}, backward {backward}, {n} seeds: l2 error of single draws (mean), of the mean over draws, ratio")
    tot_s, tot_m, tot_r = 0.0, 0.0, 0.0
    for k in keys:
        ref = a["grad." + k].double()
        single = sum(((d[k].double() - ref).norm() / ref.norm()).item() for d in draws) / n
        mean = ((sum(d[k].double() for d in draws) / n - ref).norm() / ref.norm()).item()
        tot_s += sum(((d[k].double() - ref).norm() ** 2).item() for d in draws) / n
        tot_m += ((sum(d[k].double() for d in draws) / n - ref).norm() ** 2).item()
        tot_r += (ref.norm() ** 2).item()
        print(f"   {k:42s} {single:.3e}  {mean:.3e}  {mean / single:.2f}")
    print(f"   ALL TENSORS                                {(tot_s / tot_r) ** 0.5:.3e}  {(tot_m / tot_r) ** 0.5:.3e}  {(tot_m / tot_s) ** 0.5:.2f}   (1/sqrt(n) = {n ** -0.5:.2f})")
