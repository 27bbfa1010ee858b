"""Print the gradient errors of the HIP backward vs the reference's golden gradients, per measure of
tests/test_grad_gpu.py (not a test; run on the GPU box to calibrate / re-check its thresholds)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_grad_gpu as T

overall = {}
for name in T.CASES:
    try:
        cfg, a, got, loss = T.run_case(name)
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
