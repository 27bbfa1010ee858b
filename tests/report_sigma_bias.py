"""Signed error of the density-head gradients against the reference's autograd (not a test): is the fp16 backward's
error on static_sigma / transient_sigma a bias (same sign everywhere) or noise?   python tests/report_sigma_bias.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_grad_gpu as T

for name in T.CASES:
    cfg, a, got, loss = T.run_case(name)
    out = []
    for key in sorted(a):
        if key.startswith("grad.") and ("sigma" in key) and key[5:] in got:
            g, ref = got[key[5:]].double().flatten(), a[key].double().flatten()
            along = float((g - ref) @ ref / (ref @ ref))                  # error component along the true gradient
            if ref.numel() == 1:
                out.append(f"{key[5:]}: ref {float(ref):+.4e} err {float(g - ref) / abs(float(ref)):+.2e}")
            else:
                out.append(f"{key[5:]}: along-ref {along:+.2e}, |err|/|ref| {float((g - ref).norm() / ref.norm()):.2e}")
    print(name, " | ".join(out), flush=True)
