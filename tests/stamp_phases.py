"""Diagnostic (not a test): per-phase cycle totals of the x3 fine-pass kernel from the in-kernel
s_memtime stamps of the diag library.  Run as
    NFL_LIB=$PWD/nerf_fl_amd/libnerf_fl_amd_diag.so python tests/stamp_phases.py
(build: make -C nerf_fl_amd/csrc diag)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_fl_amd
from nerf_fl_amd import NeRF, _lib, rendering as rnd
from oracle import nerfw_oracle as orc

NST = 20
NAMES = ["setup+PE", "L1", "L2", "L3", "L4", "L5", "L6", "L7", "L8", "sigma head", "final", "dir PE", "dir layer",
         "rgb head", "[consume: DMA wait]", "composite 1", "barrier", "composite 2", "exit drain", "[consume: barrier]"]
# MFMAs per 32-sample tile and phase (f16x3: 3 products)
MFMA = {"L1": 8 * 4 * 3, "L2": 384, "L3": 384, "L4": 384, "L5": 8 * 20 * 3, "L6": 384, "L7": 384, "L8": 384,
        "sigma head": 48, "final": 384, "dir layer": 4 * 18 * 3, "rgb head": 24}

dev = torch.device("cuda", 0)
R, F = 4096, 128
nerf_fl_amd.set_precision("f16x3")
m = NeRF("fine")
m.load_state_dict(orc.make_field_params(orc.FieldSpec("fine"), 12, "sharp"))
m = m.to(dev)
f = rnd._field(m, 10, 4, dev)
rays = orc.make_rays(R, 100).to(dev)
z = torch.sort(2 + 4 * torch.rand(R, F, device=dev), dim=1)[0]
noise = torch.randn(R, F, device=dev)
for _ in range(20):
    rnd._run_pass(f, rays, F, z=z, noise=noise, noise_std=1.0, white_back=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    rnd._run_pass(f, rays, F, z=z, noise=noise, noise_std=1.0, white_back=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
L = C.CDLL(_lib.LIB_PATH)
n = 256 * 4 * NST
buf = (C.c_ulonglong * n)()
assert L.nfl_debug_stamps(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, NST).astype(np.float64)
ntiles = 16                                         # 16 rays x 4 segments per workgroup / 4 waves
per = t.mean(axis=(0, 1)) / ntiles
tot = per.sum() - per[14] - per[19]      # the two consume() totals are also inside the layer phases
print(f"launch {ms:.3f} ms (stamped build); wave cycles per 32-sample tile: {tot:.0f} "
      f"(s_memtime ticks; 100 MHz-constant or shader clock per guide) -> ticks/launch/wave {t.sum(axis=2).mean():.0f}")
ideal_total = 0
for i, nm in enumerate(NAMES):
    ideal = MFMA.get(nm, 0) * 32
    ideal_total += ideal
    print(f"{nm:20s} {per[i]:9.0f}  {100 * per[i] / tot:5.1f}%   mfma-ideal {ideal:6d}  ratio {per[i] / ideal if ideal else float('nan'):5.2f}")
print(f"sum of MFMA-ideal {ideal_total}  ({100 * ideal_total / tot:.1f}% of the tile's cycles)")
print("per-wave spread of total cycles: min %.0f max %.0f" % (t.sum(axis=2).min(), t.sum(axis=2).max()))
