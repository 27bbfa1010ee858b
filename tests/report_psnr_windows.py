"""Windowed training-loss curves and validation PSNR of repeated HIP fits against the ensemble of stored reference runs
(not a test).   python tests/report_psnr_windows.py [n] [base|nerfw] [_loss=oracle] [_adam=torch] [_grad_noise=1e-3]
Single ingredients can be swapped: the oracle's autograd loss instead of the fused one, torch's Adam instead of this
package's, an extra noise floor (x max|g| per tensor) on the gradients."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_psnr_parity_gpu as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
kind = sys.argv[2] if len(sys.argv) > 2 else "base"
opts = dict(a.split("=") for a in sys.argv[3:])
if "backward" in opts:        # backward=f16x3: the three-product (fp32-class) MLP backward
    import nerf_fl_amd
    nerf_fl_amd.set_precision(backward=opts.pop("backward"))
    print("backward arithmetic:", nerf_fl_amd.rendering.get_backward_precision())
if "_grad_noise" in opts:
    opts["_grad_noise"] = float(opts["_grad_noise"])
win = 50
wmean = lambda x: np.asarray(x, np.float64)[: len(x) // win * win].reshape(-1, win).mean(1)
refs = T.reference_runs(kind)
R = np.stack([wmean(r["losses"]) for r in refs])
print("reference runs:", len(refs), "val PSNR", " ".join(f"{float(r['val_psnr']):.3f}" for r in refs))
print("ref mean curve    ", " ".join(f"{v:.5f}" for v in R.mean(0)))
print("ref rel. std (%)  ", " ".join(f"{100 * v:7.2f}" for v in R.std(0, ddof=1) / R.mean(0)))
H, P = [], []
for i in range(n):
    losses, psnr = T.fit_64_64(kind, rounding_seed=i, **opts)
    H.append(wmean(losses))
    P.append(psnr)
    print(f"hip run {i} dev (%)   ", " ".join(f"{100 * (a - b) / b:7.2f}" for a, b in zip(H[-1], R.mean(0))), f" psnr {psnr:.3f}", flush=True)
print("hip MEAN dev (%)  ", " ".join(f"{100 * (a - b) / b:7.2f}" for a, b in zip(np.stack(H).mean(0), R.mean(0))),
      f" psnr mean {np.mean(P):.3f} std {np.std(P, ddof=1) if n > 1 else 0:.3f}")
