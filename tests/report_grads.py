"""Print the gradient errors of the HIP backward vs the reference's golden gradients, per measure of
tests/test_grad_gpu.py (not a test; run on the GPU box to calibrate / re-check its thresholds)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_grad_gpu as T

if "--floor" in sys.argv:
    # How far is the REFERENCE's own fp32 autograd from the same algorithm in fp64?  (CPU, the oracle, fine depths
    # injected.)  That distance -- relu and density-threshold decisions that flip with the last bits of the forward -- is
    # the floor under any "matches the reference's gradients" statement on a fixture; the backward under test cannot be
    # expected to agree with the fp32 reference better than the fp32 reference agrees with the truth.
    import torch
    import golden_util as gu
    from oracle import nerfw_oracle as orc

    def run(name, dtype):
        torch.set_default_dtype(dtype)
        cfg, a = gu.load(name)
        (spec_c, P_c, spec_f, P_f), kw = gu.oracle_kwargs(cfg, a)
        cast = lambda t: t.to(dtype) if isinstance(t, torch.Tensor) and t.is_floating_point() else t
        P_c = {k: cast(v).requires_grad_(True) for k, v in P_c.items()}
        P_f = {k: cast(v).requires_grad_(True) for k, v in P_f.items()} if P_f is not None else None
        kw = {k: cast(v) for k, v in kw.items()}
        kw.pop("barf_epoch", None)
        kw["z_fine"] = cast(a["z_fine"]) if "z_fine" in a else None
        res = orc.render_rays(spec_c, P_c, spec_f, P_f, cast(a["rays"]), **kw)
        sum(orc.nerfw_loss(res, cast(a["target"])).values()).backward()
        out = {"coarse." + k: v.grad.double() for k, v in P_c.items()}
        out.update({"fine." + k: v.grad.double() for k, v in (P_f or {}).items() if v.grad is not None})
        return out

    for name in T.CASES:
        g32, g64 = run(name, torch.float32), run(name, torch.float64)
        rows = sorted(((g32[k] - g64[k]).abs().max().item() / g64[k].abs().max().item(), (g32[k] - g64[k]).norm().item() / g64[k].norm().item(), k)
                      for k in g64 if g64[k].abs().max() > 0)[::-1]
        print(f"== {name}: fp32 autograd vs fp64, worst tensors: " + "  ".join(f"{k}: max {m:.1e} l2 {l:.1e}" for m, l, k in rows[:3]))
    sys.exit(0)
BACKWARD = sys.argv[sys.argv.index("--backward") + 1] if "--backward" in sys.argv else "f16"
if "--seed" in sys.argv:           # the draws of the stochastic rounding (f16 / f16w): another seed, another realisation of the error
    import nerf_fl_amd
    nerf_fl_amd.set_rounding_seed(int(sys.argv[sys.argv.index("--seed") + 1]))
print("backward arithmetic:", BACKWARD)
overall = {}
for name in T.CASES:
    try:
        cfg, a, got, loss = T.run_case(name, BACKWARD)
    except Exception as e:
        import traceback
        traceback.print_exc()
        print(name, "ERROR", repr(e)[:300])
        continue
    print(f"== {name}: loss {loss:.6f} (ref {a['loss'].item():.6f})")
    rows = {}
    for measure, key, val in T.compare(cfg, a, got):
        rows.setdefault(measure, []).append((val, key))
    for measure, lst in rows.items():
        lst.sort(reverse=True)
        overall[measure] = max(overall.get(measure, (0, "")), (lst[0][0], f"{name}:{lst[0][1]}"))
        print(f"   {measure:5s} worst " + "  ".join(f"{k.split('.', 1)[1]}={v:.2e}" for v, k in lst[:4]))
print("== overall worst per measure")
for m, (v, k) in overall.items():
    print(f"   {m:5s} {v:.3e}  {k}   (threshold {T.THRESH[m][0]:.1e})")
