"""Print max abs error of the HIP path vs the golden vectors for every render fixture, with the reference's fine depths
injected (what tests/test_parity_gpu.py::test_render_at_reference_depths asserts) and end to end, for both precisions
(not a test; run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu
import gpu_util
import test_parity_gpu as T

worst_all = {}
for prec in ("f16x3", "f16"):
    for n in T.RENDER:
        cfg, a = gu.load(n)
        for inject in ((True, False) if cfg["I"] > 0 else (False,)):
            specs, kw = gu.oracle_kwargs(cfg, a)
            if inject:
                kw["z_fine"] = a["z_fine"]
            try:
                got = gpu_util.hip_render(specs, a["rays"], kw, precision=prec)
            except Exception as e:
                print(prec, n, "ERROR", repr(e)[:200])
                continue
            errs = {k: (got[k] - a["out." + k]).abs().max().item() for k in cfg["keys"]}
            w = max(errs, key=errs.get)
            tag = "reference depths" if inject else "end to end      "
            print(f"{prec:6s} {n:26s} {tag} worst {w:24s} {errs[w]:.3e}")
            if inject or cfg["I"] == 0:
                worst_all[prec] = max(worst_all.get(prec, (0, "")), (errs[w], f"{n}:{w}"))
print("worst with the reference's depths:", worst_all)
