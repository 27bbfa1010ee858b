"""Print the gradient errors of the HIP backward vs the reference's golden gradients."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_grad_gpu as T

for name in T.CASES:
    try:
        cfg, a, got, loss = T.run_case(name)
    except Exception as e:
        import traceback; traceback.print_exc()
        print(name, "ERROR", repr(e)[:300]); continue
    print(f"== {name}: loss {loss:.6f} (ref {a['loss'].item():.6f})")
    rows = sorted(T.compare(cfg, a, got), key=lambda r: -(r[1] / (r[2] + 1e-12)))
    for key, err, ref in rows[:8]:
        print(f"   {key:52s} err {err:.3e}  ref_max {ref:.3e}  rel {err / (ref + 1e-12):.2e}")
