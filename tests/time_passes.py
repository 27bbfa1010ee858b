"""Time the individual kernels of one cfg-2 train step (run on the GPU box; not a test)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_fl_amd
from nerf_fl_amd import NeRF, PosEmbedding, _lib
from nerf_fl_amd import rendering as rnd
from oracle import nerfw_oracle as orc

dev = torch.device("cuda", 0)
R, S, F = 4096, 64, 128
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
nerf_fl_amd.set_precision(prec)
m = NeRF("fine")
m.load_state_dict(orc.make_field_params(orc.FieldSpec("fine"), 12, "sharp"))
m = m.to(dev)
f = rnd._field(m, 10, 4, dev)
BWD = sys.argv[sys.argv.index("--backward") + 1] if "--backward" in sys.argv else "f16"      # f16 | f16x3
nerf_fl_amd.set_precision(backward=BWD)
BP = rnd._BPREC[BWD]
bp = f.ensure_bwd_packed(False, BP)
rays = orc.make_rays(R, 100).to(dev)
z = torch.sort(2 + 4 * torch.rand(R, F, device=dev), dim=1)[0]
noise = torch.randn(R, F, device=dev)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"precision {prec}; fine pass {R} rays x {F} samples")
print("fwd  (inference)      %.3f ms" % timeit(lambda: rnd._run_pass(f, rays, F, z=z, noise=noise, noise_std=1.0, white_back=True)))
if prec == "f16x3":
    out = {}
    def fwd_stash():
        out.update(rnd._run_pass(f, rays, F, z=z, noise=noise, noise_std=1.0, white_back=True, stash=True, bprec=BP))
    print("fwd  (training stash) %.3f ms" % timeit(fwd_stash))
    st = dict(z=z, field_raw=out["field_raw"], act=out["act_stash"], noise=noise, use_t=False, n=F)
    cfg = dict(noise_std=1.0, white_back=True)
    keys = ["weights_fine", "opacity_fine", "rgb_fine", "depth_fine"]
    grads = [None, None, torch.randn(R, 3, device=dev) * 1e-3, None]
    L = _lib.lib()
    head = torch.empty(R * F, 9, device=dev)
    grad_stash = torch.empty(L.nfl_grad_stash_bytes(C.byref(f.desc), R, F, BP), dtype=torch.uint8, device=dev)
    ca = _lib.CompBwdArgs()
    ca.d_field_raw, ca.d_z, ca.d_noise = rnd._ptr(st["field_raw"]), rnd._ptr(z), rnd._ptr(noise)
    ca.noise_std, ca.n_rays, ca.n_samples, ca.use_transient, ca.white_back = 1.0, R, F, 0, 1
    ca.g_rgb = rnd._ptr(grads[2])
    ca.d_head_grads = rnd._ptr(head)
    gmax = torch.zeros(1024, device=dev)
    ca.d_gmax = rnd._ptr(gmax)
    print("composite backward    %.3f ms" % timeit(lambda: _lib.check(L.nfl_composite_backward(C.byref(ca), rnd._stream()), "cb")))
    da = _lib.DgradArgs()
    da.d_head_grads, da.d_act_stash, da.d_grad_stash = rnd._ptr(head), rnd._ptr(st["act"]), rnd._ptr(grad_stash)
    da.n_rays, da.n_samples, da.use_transient = R, F, 0
    da.d_gmax = rnd._ptr(gmax)
    print("dgrad                 %.3f ms" % timeit(lambda: _lib.check(L.nfl_mlp_dgrad(bp['h'], rnd._ptr(bp['d']), rnd._ptr(bp['packed']), C.byref(da), rnd._stream()), "dg")))
    plist = f.param_list()
    arena = torch.zeros(sum(w.numel() + b.numel() for _, w, b in plist), device=dev)
    fg = _lib.FieldGrads()
    off = 0
    for i, w, b in plist:
        fg.weight[i] = arena[off:off + w.numel()].data_ptr(); off += w.numel()
        fg.bias[i] = arena[off:off + b.numel()].data_ptr(); off += b.numel()
    h_wp, d_wp = f.wgrad_plan(False)
    fpar, _keep = f._field_params()
    scratch = torch.empty(L.nfl_wgrad_scratch_bytes() // 4, device=dev)
    print("wgrad                 %.3f ms" % timeit(lambda: _lib.check(L.nfl_mlp_wgrad(h_wp, rnd._ptr(d_wp), rnd._ptr(st["act"]), rnd._ptr(grad_stash), rnd._ptr(gmax), R, F, BP, C.byref(fpar), rnd._ptr(scratch), C.byref(fg), rnd._stream()), "wg")))
    print("act stash %.2f GB, grad stash %.2f GB" % (st["act"].numel() / 1e9, grad_stash.numel() / 1e9))
