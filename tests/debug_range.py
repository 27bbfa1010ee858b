import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import test_range_gpu as T
from nerf_fl_amd import rendering
for K in (2.0 ** 15, 2.0 ** 17, 2.0 ** 20):
    got, exp = T._run(K)
    print("K", K, {k: (int(torch.isnan(v).sum()), int(torch.isinf(v).sum()), float((v - exp[k]).abs().nan_to_num(0).max())) for k, v in got.items()})
    print("  status words:", {k: int(w.item()) for k, w in rendering._status_words.items()})
    for w in rendering._status_words.values():
        w.zero_()
