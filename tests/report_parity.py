"""Print max abs error of the HIP path vs the golden vectors for every render fixture
and both precisions (not a test; run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu
import gpu_util

names = [n for n in gu.golden_names("g") if n[:2] in ("g4", "g5", "g6", "g7", "g8", "g9")
         or n.startswith(("g10", "g12_stoch_base", "g12_stoch_nerfw", "g13"))]
for prec in ("f16x3", "f16"):
    for n in names:
        cfg, a = gu.load(n)
        specs, kw = gu.oracle_kwargs(cfg, a)
        try:
            got = gpu_util.hip_render(specs, a["rays"], kw, precision=prec)
        except Exception as e:
            print(prec, n, "ERROR", repr(e)[:200]); continue
        errs = {k: (got[k] - a["out." + k]).abs().max().item() for k in cfg["keys"]}
        w = max(errs, key=errs.get)
        print(f"{prec:6s} {n:28s} worst {w:24s} {errs[w]:.3e}   keys_ok={list(got.keys()) == cfg['keys']}")
