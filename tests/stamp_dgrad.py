"""Diagnostic (not a test): cycles of the dgrad kernel spent in the ring's consume() (DMA wait / barrier), from the
in-kernel s_memtime stamps of a diag library built with  make -C nerf_fl_amd/csrc diag DIAGTU=nfl_dgrad DIAGNAME=_dg
    NFL_LIB=$PWD/nerf_fl_amd/libnerf_fl_amd_diag_dg.so python tests/stamp_dgrad.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_fl_amd
from nerf_fl_amd import NeRF, _lib, rendering as rnd
from oracle import nerfw_oracle as orc

dev = torch.device("cuda", 0)
R, F = 4096, 128
m = NeRF("fine")
m.load_state_dict(orc.make_field_params(orc.FieldSpec("fine"), 12, "sharp"))
m = m.to(dev)
f = rnd._field(m, 10, 4, dev)
bp = f.ensure_bwd_packed(False)
rays = orc.make_rays(R, 100).to(dev)
z = torch.sort(2 + 4 * torch.rand(R, F, device=dev), dim=1)[0]
noise = torch.randn(R, F, device=dev)
out = rnd._run_pass(f, rays, F, z=z, noise=noise, noise_std=1.0, white_back=True, stash=True, field_raw=True)
L = _lib.lib()
head = torch.randn(R * F, 9, device=dev) * 1e-3
gmax = torch.full((1024,), 4e-3, device=dev)
grad_stash = torch.empty(L.nfl_grad_stash_bytes(C.byref(f.desc), R, F, _lib.NFL_PREC_F16), dtype=torch.uint8, device=dev)
da = _lib.DgradArgs()
da.d_head_grads, da.d_act_stash, da.d_grad_stash = rnd._ptr(head), rnd._ptr(out["act_stash"]), rnd._ptr(grad_stash)
da.n_rays, da.n_samples, da.use_transient, da.d_gmax = R, F, 0, rnd._ptr(gmax)
run = lambda: _lib.check(L.nfl_mlp_dgrad(bp["h"], rnd._ptr(bp["d"]), rnd._ptr(bp["packed"]), C.byref(da), rnd._stream()), "dg")
for _ in range(10):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
H = C.CDLL(_lib.LIB_PATH)
n = 256 * 4 * 20
buf = (C.c_ulonglong * n)()
assert H.nfl_debug_stamps(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, 20).astype(np.float64)
tot, wait, bar, ncons, ntiles = (t[:, :, i].mean() for i in range(5))
print(f"launch {e0.elapsed_time(e1) / 10:.3f} ms; per wave: {tot:.0f} cycles, {ncons:.0f} consumes over {ntiles:.0f} tile passes")
print(f"per consume: total {tot / ncons:.0f}  DMA/store wait {wait / ncons:.0f}  barrier {bar / ncons:.0f}  rest {(tot - wait - bar) / ncons:.0f}")
