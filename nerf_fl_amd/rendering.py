"""render_rays on MI355X -- drop-in for the reference's models/rendering.py.

Same call boundary as the reference (models/rendering.py:49-62): the same
positional argument order (all three reference callers pass the first eleven
positionally: train.py:100-111, eval.py:91-103), the same recognised kwargs
(`view_dir`, `a_embedded`, `t_embedded`, `output_transient`; unknown ones are
ignored), and the same result keys in the same insertion order (SURVEY.md
appendix B).  Everything between the arguments and the result dict runs in the
HIP library behind include/nerf_fl_amd.h:

    nfl_pack_field   (only when parameters changed)
    nfl_render_pass  coarse   -> weights/opacity/rgb/depth _coarse
    nfl_sample_pdf            -> sorted fine depths
    nfl_render_pass  fine     -> ..._fine, transient outputs

Randomness: the reference draws rand_like (jitter), randn_like (density noise,
even when noise_std == 0), rand (importance u), randn_like (fine noise) in that
order.  The same draws are made here with torch's
generator on the rays' device, unless the caller injects them through the
build-defined kwargs `perturb_rand`, `noise_coarse`, `u`, `noise_fine`; `z_fine` (R, N_samples + N_importance)
injects the sorted fine depths themselves (the importance sampler is then skipped).

Numerical range: the fused MLP multiplies fp16 operands (split hi+lo), so activations and weights must stay
within fp16's range (|x| <= 65504); beyond it the affected rays come out as NaN where the fp32 reference stays
finite.  Every pass records that in a device status word; `check_status()` (or the build-defined kwarg
`check_finite=True`) turns it into a FloatingPointError.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib

__all__ = ["render_rays", "set_precision", "get_precision", "get_backward_precision", "check_status", "CameraRays"]

_PREC = {"f16x3": _lib.NFL_PREC_F16X3, "f16": _lib.NFL_PREC_F16}
_BPREC = dict(_PREC, f16w=_lib.NFL_PREC_F16W)
_precision = os.environ.get("NERF_FL_AMD_PREC", "f16x3")
_backward = os.environ.get("NERF_FL_AMD_BWD", "f16")


def set_precision(name=None, backward=None):
    """Forward arithmetic `name`: 'f16x3' (default; fp16 MFMA with split operands, 3 products, fp32-class accuracy) or
    'f16' (single product; fast, inference only).
    `backward`: arithmetic of the MLP part of the hand-written backward (dgrad + wgrad):
      'f16'    (default) single fp16 products on fp16 stashes under a loss scale, fastest.  The fp16 roundings of the
               gradients and of the transposed weights the gradient chain multiplies by are DRAWN (stochastic rounding:
               zero-mean errors, see set_rounding_seed): a single step's gradient is within ~6e-3 of fp32 autograd, long
               training curves follow the reference's to <= 1 % (profiles/r03_psnr_backward_attribution.txt);
      'f16w'   the same, but the chain reads hi + lo weight fragments (two products, exact weights): per-step gradients
               within ~3e-3, training curves inside the reference's own run-to-run scatter; +10 % step time;
      'f16x3'  split operands hi + lo everywhere, 3 products, hi + lo activation / gradient stashes: fp32-class gradients,
               the reference's precision class, at about twice the backward's HBM traffic and three times its matrix work."""
    global _precision, _backward
    if name is not None:
        if name not in _PREC:
            raise ValueError(f"precision must be one of {sorted(_PREC)}")
        _precision = name
    if backward is not None:
        if backward not in _BPREC:
            raise ValueError(f"backward must be one of {sorted(_BPREC)}")
        _backward = backward


def get_precision():
    return _precision


_rounding_seed = 0


def set_rounding_seed(seed):
    """Seed of the stochastic rounding in the 'f16' / 'f16w' backward (see set_precision).  The draws are a function of
    (seed, work-item, the pass's gradient maximum): a fit repeated with the same seed sees the same draws as long as its
    data are bit-identical; independent repetitions of a fit should use different seeds."""
    global _rounding_seed
    _rounding_seed = int(seed) & 0xFFFFFFFF


def get_backward_precision():
    return _backward


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class CameraRays:
    """Stand-in for a `rays` matrix whose rows are the pixels [start, start + count) of an H x W frame seen through a
    pinhole camera (pose `c2w` (3|4, 4), intrinsics `K` (3, 3), bounds near / far): pass it to render_rays instead of the
    (N_rays, 8) tensor and the rays are generated inside the render kernel's prologue (reference
    datasets/ray_utils.py:5-55 + the near/far columns of the datasets) -- an eval loop then moves 12 pose floats per
    frame instead of 32 B per ray.  Inference only.  `nerf_fl_amd.eval.frame_rays` materialises the same rows."""

    def __init__(self, c2w, K, H, W, near, far, device, start=0, count=None):
        count = H * W - start if count is None else int(count)
        if start < 0 or count < 0 or start + count > H * W:
            raise ValueError("pixel range outside the frame")
        c2w = torch.as_tensor(c2w, dtype=torch.float32).cpu()[:3, :4].contiguous()
        K = torch.as_tensor(K, dtype=torch.float32).cpu()
        self.device = torch.device(device)
        self.H, self.W, self.start, self.count = int(H), int(W), int(start), count
        self.shape = (count, 8)
        self.requires_grad = False
        self.cam = _lib.Camera()
        for k, v in enumerate(c2w.reshape(-1).tolist()):
            self.cam.c2w[k] = v
        self.cam.fx, self.cam.fy, self.cam.cx, self.cam.cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
        self.cam.width, self.cam.reserved, self.cam.pix0 = int(W), 0, int(start)
        self.cam.near, self.cam.far = float(near), float(far)

    def slice(self, lo, hi):
        """Rows [lo, hi) as another CameraRays (what `rays[lo:hi]` is for a tensor)."""
        out = CameraRays.__new__(CameraRays)
        out.__dict__.update(self.__dict__)
        hi = min(hi, self.count)
        out.start, out.count, out.shape = self.start + lo, hi - lo, (hi - lo, 8)
        out.cam = _lib.Camera.from_buffer_copy(self.cam)
        out.cam.pix0 = self.start + lo
        out.d_cam = None
        return out

    d_cam = None       # device copy of the camera struct (to_device): the pass then reads the camera through a pointer

    def to_device(self):
        """Keep the camera in device memory (C ABI nfl_pass_args::d_cam): a render pass captured in a HIP graph then
        renders whatever `load()` last put there, instead of the camera frozen into the captured launch."""
        if self.d_cam is None:
            self.d_cam = torch.zeros(C.sizeof(_lib.Camera), dtype=torch.uint8, device=self.device)
        self.d_cam.copy_(torch.frombuffer(bytearray(bytes(self.cam)), dtype=torch.uint8))
        return self

    def load(self, other):
        """Take over `other`'s camera and pixel range, keeping this object's device buffer.  (A captured launch renders
        its fixed number of rays whatever the range holds: rows past `other.count` are the pixels that follow in row-major
        order -- beyond the last row of the frame they are simply rays below it; nothing is indexed by pixel -- and the
        caller drops them.)"""
        self.cam = _lib.Camera.from_buffer_copy(other.cam)
        self.H, self.W, self.start = other.H, other.W, other.start
        self.to_device()
        return self


_status_words = {}


def _status_word(dev):
    """One int32 per device that every render pass ORs NFL_STATUS_* bits into (include/nerf_fl_amd.h: d_status)."""
    k = str(dev)
    if k not in _status_words:
        _status_words[k] = torch.zeros(1, dtype=torch.int32, device=dev)
    return _status_words[k]


def check_status(device=None):
    """Synchronise and raise FloatingPointError if a render pass since the last check produced a non-finite
    per-ray output: the MLP runs on fp16 operands, so an activation or weight beyond |x| = 65504 (which the fp32
    reference would carry) becomes NaN here -- the documented range limit of this build.
    The status word is cleared."""
    for k, w in list(_status_words.items()):
        if device is not None and k != str(torch.device(device)):
            continue
        bits = int(w.item())
        if bits:
            w.zero_()
            what = []
            if bits & _lib.NFL_STATUS_RANGE:
                what.append("a weight or an activation exceeded fp16's range (|x| > 65504) inside the fused MLP")
            if bits & _lib.NFL_STATUS_NONFINITE:
                what.append("a composited per-ray output was not finite")
            raise FloatingPointError(f"nerf_fl_amd: render pass on {k}: " + "; ".join(what)
                                     + " (see INTEGRATION.md, 'Numerical range')")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t, name, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"nerf_fl_amd.render_rays: `{name}` must be a CUDA/HIP tensor (this build has no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"`{name}` must be float32, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"`{name}` has shape {tuple(t.shape)}, expected {tuple(shape)}")
    t = t.detach().contiguous()
    if t.data_ptr() % 16:          # the kernels use 16-byte vector loads
        t = t.clone()
    return t


class _PackedField:
    """Device-side state of one nn.Module field: plan + packed weight stream."""

    def __init__(self, model, n_emb_xyz, n_emb_dir, prec, device):
        L = _lib.lib()
        self.model_ref = weakref.ref(model)
        self.device = device
        self.desc = _lib.FieldDesc(
            n_emb_xyz=n_emb_xyz, n_emb_dir=n_emb_dir,
            encode_appearance=int(bool(model.encode_appearance)), n_a=int(getattr(model, "in_channels_a", 48) or 48),
            encode_transient=int(bool(model.encode_transient)), n_tau=int(getattr(model, "in_channels_t", 16)),
            beta_min=float(model.beta_min), reserved=0)
        if model.in_channels_xyz != 6 * n_emb_xyz + 3 or model.in_channels_dir != 6 * n_emb_dir + 3:
            raise ValueError("embedding widths do not match the model's in_channels_xyz / in_channels_dir")
        nbytes = L.nfl_plan_bytes(C.byref(self.desc))
        self.h_plan = C.create_string_buffer(nbytes)
        _lib.check(L.nfl_plan_build(C.byref(self.desc), prec, self.h_plan, nbytes), "nfl_plan_build")
        self.d_plan = torch.frombuffer(bytearray(self.h_plan.raw), dtype=torch.uint8).to(device)
        self.packed_bytes = L.nfl_packed_bytes(C.byref(self.desc), prec)
        self.packed = torch.empty(self.packed_bytes, dtype=torch.uint8, device=device)
        self.key = None
        # dgrad stream (transposed weights, fp16, one product); built lazily on the first training forward
        self._slots = None        # (submodule, parameter slot, qualified name) of every parameter: _named()
        self.fold = None          # folded copies of dir_encoding.0 / transient_encoding.0 (weight, bias) for the packer
        self.bplans = {}          # (rays_grad, backward precision) -> dict(h, d, packed, nbytes, key)
        self.wplans = {}          # use_transient -> (host blob, device copy) of the wgrad job list
        self.wg_scratch = None    # composition scratch of nfl_mlp_wgrad when the gradients go to a caller-owned GradArena

    def _named(self):
        """name -> nn.Parameter, as dict(model.named_parameters()) but without walking the module tree on every call (73 us
        per walk, ten walks per training step: at the README batch of 1024 rays that was a quarter of the eager step's host
        time).  The (submodule, slot) pairs are collected once; the Parameter objects are read through them, so a
        parameter that is re-assigned is still seen."""
        if self._slots is None:
            self._slots = [(mod, pn, (prefix + "." if prefix else "") + pn)
                           for prefix, mod in self.model_ref().named_modules() for pn in mod._parameters]
        return {q: mod._parameters[pn] for mod, pn, q in self._slots if mod._parameters[pn] is not None}

    def wgrad_plan(self, use_t):
        if use_t not in self.wplans:
            L = _lib.lib()
            n = L.nfl_wgrad_plan_bytes()
            h = C.create_string_buffer(n)
            _lib.check(L.nfl_wgrad_plan_build(C.byref(self.desc), int(use_t), h, n), "nfl_wgrad_plan_build")
            self.wplans[use_t] = (h, torch.frombuffer(bytearray(h.raw), dtype=torch.uint8).to(self.device))
        return self.wplans[use_t]

    def param_list(self):
        """(layer index, weight, bias) for every layer this field has, in NFL_P_* order."""
        params = self._named()
        out = []
        for i, name in enumerate(_lib.LAYER_NAMES):
            if name + ".weight" in params:
                out.append((i, params[name + ".weight"], params[name + ".bias"]))
        return out

    def bwd_plan(self, rays_grad=False, bprec=None):
        """Plan + buffer of the dgrad stream (transposed weights, fp16 -- hi + lo fragments for the three-product backward);
        packed by ensure_bwd_packed / _pack_streams."""
        L = _lib.lib()
        bprec = _BPREC[_backward] if bprec is None else bprec
        rg = (int(bool(rays_grad)), bprec)
        if rg not in self.bplans:
            nbytes = L.nfl_plan_bytes(C.byref(self.desc))
            h = C.create_string_buffer(nbytes)
            _lib.check(L.nfl_bwd_plan_build(C.byref(self.desc), rg[0], bprec, h, nbytes), "nfl_bwd_plan_build")
            pb = L.nfl_bwd_packed_bytes(C.byref(self.desc), rg[0], bprec)
            self.bplans[rg] = dict(h=h, d=torch.frombuffer(bytearray(h.raw), dtype=torch.uint8).to(self.device),
                                   packed=torch.empty(pb, dtype=torch.uint8, device=self.device), nbytes=pb, key=None)
        return self.bplans[rg]

    def ensure_bwd_packed(self, rays_grad=False, bprec=None):
        """dgrad stream for the current parameters (the forward stream must be current: self.key)."""
        L = _lib.lib()
        bp = self.bwd_plan(rays_grad, bprec)
        if bp["key"] != self.key:
            fp, _keep = self._pack_params()
            _lib.check(L.nfl_pack_field(bp["h"], _ptr(bp["d"]), C.byref(fp), _ptr(bp["packed"]), bp["nbytes"],
                                        C.c_void_p(0), _stream()), "nfl_pack_field(bwd)")
            bp["key"] = self.key
        return bp

    def _field_params(self):
        params = self._named()
        fp = _lib.FieldParams()
        keep = []
        for i, name in enumerate(_lib.LAYER_NAMES):
            w, b = params.get(name + ".weight"), params.get(name + ".bias")
            if w is None:
                fp.weight[i] = None
                fp.bias[i] = None
                continue
            w, b = _f32c(w, name + ".weight"), _f32c(b, name + ".bias")
            keep += [w, b]
            fp.weight[i] = w.data_ptr()
            fp.bias[i] = b.data_ptr()
        return fp, keep

    def _pack_params(self):
        """The parameters as the PACKED streams see them: xyz_encoding_final folded into the first 256 input columns of
        dir_encoding.0 / transient_encoding.0 (include/nerf_fl_amd.h: nfl_compose_forward).  One launch + two (four) small
        copies per re-pack; the gradients still go to the original parameters (nfl_mlp_wgrad composes them)."""
        fp, keep = self._field_params()
        params = self._named()
        has_t = "transient_encoding.0.weight" in params
        names = ["dir_encoding.0"] + (["transient_encoding.0"] if has_t else [])
        if self.fold is None:
            self.fold = {n: (torch.empty_like(params[n + ".weight"], dtype=torch.float32).contiguous(),
                             torch.empty_like(params[n + ".bias"], dtype=torch.float32).contiguous()) for n in names}
        for n in names:
            # the side columns ride along; the first 256 are overwritten.  An elementwise KERNEL (x * 1 is exact), not copy_: a
            # device-to-device copy_ is a hipMemcpyAsync, i.e. a memcpy NODE inside a captured step, and nodes of that kind were seen
            # to take effect out of order under a second process on the same GPU
            torch.mul(params[n + ".weight"].detach(), 1.0, out=self.fold[n][0])
        wd, bd = self.fold["dir_encoding.0"]
        wt, bt = self.fold["transient_encoding.0"] if has_t else (None, None)
        n_side = 6 * int(self.desc.n_emb_dir) + 3 + int(self.desc.n_a if self.desc.encode_appearance else 0)
        _lib.check(_lib.lib().nfl_compose_forward(C.byref(fp), int(has_t), n_side,
                                                  int(self.desc.n_tau), _ptr(wd), _ptr(bd), _ptr(wt), _ptr(bt), _stream()),
                   "nfl_compose_forward")
        for n in names:
            i = _lib.LAYER_NAMES.index(n)
            fp.weight[i], fp.bias[i] = self.fold[n][0].data_ptr(), self.fold[n][1].data_ptr()
        return fp, keep

    def current_key(self):
        return tuple((n, p.data_ptr(), p._version) for n, p in self._named().items())

    def ensure_packed(self):
        key = self.current_key()
        if key == self.key:
            return
        fp, _keep = self._pack_params()
        _lib.check(_lib.lib().nfl_pack_field(self.h_plan, _ptr(self.d_plan), C.byref(fp), _ptr(self.packed),
                                             self.packed_bytes, _ptr(_status_word(self.device)), _stream()),
                   "nfl_pack_field")
        self.key = key


_fields = weakref.WeakKeyDictionary()
_lin_cache = {}


def _field(model, n_emb_xyz, n_emb_dir, device, pack=True):
    prec = _PREC[_precision]
    slot = _fields.setdefault(model, {})
    k = (prec, str(device), n_emb_xyz, n_emb_dir)
    if k not in slot:
        slot[k] = _PackedField(model, n_emb_xyz, n_emb_dir, prec, device)
    f = slot[k]
    if pack:
        f.ensure_packed()
    return f


def _pack_streams(fields, bwd, rays_grad):
    """Bring the forward (and, for a training call, the dgrad) weight streams of `fields` up to date with ONE
    nfl_pack_fields launch: after an optimizer step all four streams of a coarse + fine pair are stale."""
    jobs, done, keep = [], [], []
    for f in fields:
        if f is None or any(f is g for g, _ in done):
            continue
        key = f.current_key()
        done.append((f, key))
        stale_fwd = key != f.key
        bp = f.bwd_plan(rays_grad) if bwd else None
        stale_bwd = bp is not None and bp["key"] != key
        if not (stale_fwd or stale_bwd):
            continue
        fp, tensors = f._pack_params()
        keep += [fp, tensors]
        if stale_fwd:
            jobs.append((f, None, key, _lib.PackJob(C.cast(f.h_plan, C.c_void_p), _ptr(f.d_plan), C.pointer(fp), _ptr(f.packed),
                                                    f.packed_bytes, _ptr(_status_word(f.device)))))
        if stale_bwd:
            jobs.append((f, bp, key, _lib.PackJob(C.cast(bp["h"], C.c_void_p), _ptr(bp["d"]), C.pointer(fp), _ptr(bp["packed"]),
                                                  bp["nbytes"], C.c_void_p(0))))
    for i in range(0, len(jobs), _lib.NFL_PACK_MAX_JOBS):
        part = jobs[i:i + _lib.NFL_PACK_MAX_JOBS]
        arr = (_lib.PackJob * len(part))(*[j[3] for j in part])
        _lib.check(_lib.lib().nfl_pack_fields(len(part), arr, _stream()), "nfl_pack_fields")
    for f, bp, key, _ in jobs:
        if bp is None:
            f.key = key
        else:
            bp["key"] = key


def field_forward(model, x, sigma_only=False, output_transient=True):
    """NeRF.forward on an already-encoded (B, C) matrix (reference models/nerf.py:153-212) through the
    fused HIP kernel.  Inference only: the result carries no autograd graph (training goes through
    render_rays, whose backward is fused too)."""
    if not x.is_cuda:
        raise RuntimeError("nerf_fl_amd: NeRF.forward needs a ROCm device tensor (there is no CPU fallback)")
    if x.requires_grad and torch.is_grad_enabled():
        raise RuntimeError("nerf_fl_amd: NeRF.forward is inference-only; differentiate through render_rays")
    if x.dim() != 2:
        raise ValueError("x must be (B, C)")
    n_xyz, n_dir = (model.in_channels_xyz - 3) // 6, (model.in_channels_dir - 3) // 6
    f = _field(model, n_xyz, n_dir, x.device)
    use_t = bool(output_transient) and not sigma_only
    if use_t and not model.encode_transient:
        raise ValueError("output_transient=True needs a model built with encode_transient")
    x = x.detach().to(torch.float32).contiguous()
    B = x.shape[0]
    raw = torch.empty(B, 9, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().nfl_field_forward(f.h_plan, _ptr(f.d_plan), _ptr(f.packed), _ptr(x) if B else None, B,
                                            x.shape[1], int(bool(sigma_only)), int(use_t), _ptr(raw) if B else None,
                                            _stream()) if B else 0, "nfl_field_forward")
    if sigma_only:
        return raw[:, 3:4].contiguous()
    return raw if use_t else raw[:, :4].contiguous()


def posenc(x, n_freqs, weights=None):
    """PosEmbedding / BarfPosEmbedding.forward (reference models/nerf.py:19-32, 61-77) on the device."""
    if not x.is_cuda:
        raise RuntimeError("nerf_fl_amd: PosEmbedding.forward needs a ROCm device tensor (there is no CPU fallback)")
    if x.requires_grad and torch.is_grad_enabled():
        raise RuntimeError("nerf_fl_amd: PosEmbedding.forward is inference-only; differentiate through render_rays")
    if x.shape[-1] != 3:
        raise ValueError("x must be (..., 3)")
    lead = x.shape[:-1]
    xf = x.detach().to(torch.float32).reshape(-1, 3).contiguous()
    n = xf.shape[0]
    out = torch.empty(n, 6 * n_freqs + 3, dtype=torch.float32, device=x.device)
    w = None if weights is None else weights.to(device=x.device, dtype=torch.float32).contiguous()
    if n:
        _lib.check(_lib.lib().nfl_posenc(_ptr(xf), n, n_freqs, _ptr(w) if w is not None else None, _ptr(out), _stream()),
                   "nfl_posenc")
    return out.reshape(*lead, 6 * n_freqs + 3)


def _plain_embedding(e):
    """An nn.Embedding whose backward is a plain dense scatter-add of the looked-up rows' gradients."""
    return (isinstance(e, torch.nn.Embedding) and e.padding_idx is None and e.max_norm is None
            and not e.scale_grad_by_freq and not e.sparse and e.weight.dtype == torch.float32 and e.weight.is_contiguous())


def _linspace(n, device):
    k = (n, str(device))
    if k not in _lin_cache:
        _lin_cache[k] = torch.linspace(0, 1, n, device=device)
    return _lin_cache[k]


def _n_freqs(emb):
    n = getattr(emb, "N_freqs", None)
    return int(n) if n is not None else int(len(emb.freqs))


def _run_pass(field, rays, n_samples, *, z=None, lin=None, perturb_rand=None, perturb=0.0, use_disp=False,
              noise=None, noise_std=0.0, a_emb=None, t_emb=None, view_dir=None, sigma_only=False,
              white_back=False, test_extras=False, want_rgb=True, want_z=False, field_raw=False, stash=False,
              pe_w_xyz=None, pe_w_dir=None, loss=None, loss_slot=0, bprec=_lib.NFL_PREC_F16):
    R = rays.shape[0]
    dev = rays.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    use_t = t_emb is not None and not sigma_only
    out = {"weights": new(R, n_samples), "opacity": new(R)}
    if want_rgb and not sigma_only:
        out["rgb"] = new(R, 3)
        out["depth"] = new(R)
    if use_t:
        out.update(transient_sigmas=new(R, n_samples), beta=new(R), rgb_static=new(R, 3), rgb_transient=new(R, 3))
        if test_extras:
            out.update(rgb_static_only=new(R, 3), depth_static_only=new(R),
                       rgb_transient_only=new(R, 3), depth_transient_only=new(R))
    if want_z:
        out["z"] = new(R, n_samples)
    if field_raw or stash:
        out["field_raw"] = torch.empty(R * n_samples, 9, dtype=torch.float32, device=dev)
    if stash:
        nb = _lib.lib().nfl_act_stash_bytes(C.byref(field.desc), R, n_samples, bprec)
        out["act_stash"] = torch.empty(nb, dtype=torch.uint8, device=dev)
    a = _lib.PassArgs()
    if isinstance(rays, CameraRays):
        a.d_rays, a.h_cam = C.c_void_p(0), C.pointer(rays.cam)
        a.d_cam = _ptr(rays.d_cam)
    else:
        a.d_rays = _ptr(rays)
    a.d_view_dir = _ptr(view_dir)
    a.n_rays, a.n_samples = R, n_samples
    a.d_z, a.d_lin, a.d_perturb_rand = _ptr(z), _ptr(lin), _ptr(perturb_rand)
    a.perturb, a.use_disp = float(perturb), int(bool(use_disp))
    a.d_z_out = _ptr(out.get("z"))
    a.d_noise, a.noise_std = _ptr(noise), float(noise_std)
    a.d_a_emb, a.d_t_emb = _ptr(a_emb), _ptr(t_emb if use_t else None)
    a.sigma_only, a.white_back, a.test_extras = int(sigma_only), int(bool(white_back)), int(bool(test_extras and use_t))
    a.d_weights, a.d_opacity = _ptr(out["weights"]), _ptr(out["opacity"])
    a.d_rgb, a.d_depth = _ptr(out.get("rgb")), _ptr(out.get("depth"))
    a.d_transient_sigmas, a.d_beta = _ptr(out.get("transient_sigmas")), _ptr(out.get("beta"))
    a.d_rgb_static, a.d_rgb_transient = _ptr(out.get("rgb_static")), _ptr(out.get("rgb_transient"))
    a.d_rgb_static_only, a.d_depth_static_only = _ptr(out.get("rgb_static_only")), _ptr(out.get("depth_static_only"))
    a.d_rgb_transient_only = _ptr(out.get("rgb_transient_only"))
    a.d_depth_transient_only = _ptr(out.get("depth_transient_only"))
    a.d_field_raw = _ptr(out.get("field_raw"))
    a.d_act_stash = _ptr(out.get("act_stash"))
    a.stash_split = int(stash and bprec == _lib.NFL_PREC_F16X3)      # hi + lo records for the three-product backward
    a.d_pe_w_xyz, a.d_pe_w_dir = _ptr(pe_w_xyz), _ptr(pe_w_dir)
    if loss is not None:        # NerfWLoss fused into the per-ray epilogue (include/nerf_fl_amd.h: d_loss_target)
        out["seed_rgb"] = new(R, 3)
        if use_t:
            out["seed_beta"] = new(R)
        a.d_loss_target, a.d_losses = _ptr(loss["target"]), _ptr(loss["losses"])
        a.d_seed_rgb, a.d_seed_beta = _ptr(out["seed_rgb"]), _ptr(out.get("seed_beta"))
        a.loss_coef, a.lambda_u, a.loss_slot = float(loss["coef"]), float(loss["lambda_u"]), int(loss_slot)
    a.d_status = _ptr(_status_word(dev))
    _lib.check(_lib.lib().nfl_render_pass(field.h_plan, _ptr(field.d_plan), _ptr(field.packed), C.byref(a), _stream()),
               "nfl_render_pass")
    return out


def _forward(cfg, rays, a_emb, t_emb, train):
    """Both passes + hierarchical sampling.  Returns (result dict in the reference's key
    order, state saved for the backward or None)."""
    f_c, f_f = cfg["f_c"], cfg["f_f"]
    R, dev, S, I = rays.shape[0], rays.device, cfg["S"], cfg["I"]
    test_time, raw = cfg["test_time"], cfg["raw"]
    result, saved = {}, {}
    oc = _run_pass(f_c, rays, S, lin=_linspace(S, dev), perturb_rand=cfg["perturb_rand"], perturb=cfg["perturb"],
                   use_disp=cfg["use_disp"], noise=cfg["noise_c"], noise_std=cfg["noise_std"],
                   view_dir=cfg["view_dir"], sigma_only=test_time, white_back=cfg["white_back"],
                   want_z=I > 0 or train, field_raw=raw, stash=train, pe_w_xyz=cfg["pe_w_xyz"], pe_w_dir=cfg["pe_w_dir"],
                   loss=cfg["loss"] if train else None, loss_slot=0, bprec=cfg["bprec"])
    result["weights_coarse"] = oc["weights"]
    result["opacity_coarse"] = oc["opacity"]
    if not test_time:
        result["rgb_coarse"] = oc["rgb"]
        result["depth_coarse"] = oc["depth"]
    if raw:
        result["_field_raw_coarse"] = oc["field_raw"]
    if train:
        saved["coarse"] = dict(z=oc["z"], field_raw=oc["field_raw"], act=oc["act_stash"], noise=cfg["noise_c"],
                               use_t=False, n=S, seed_rgb=oc.get("seed_rgb"), seed_beta=None)
    if I > 0:
        F = S + I
        if cfg["z_fine"] is not None:        # injected fine depths (build-defined kwarg `z_fine`): the sampler is skipped
            z_fine = cfg["z_fine"]
        else:
            z_fine = torch.empty(R, F, dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().nfl_sample_pdf(_ptr(oc["z"]), _ptr(oc["weights"]), _ptr(cfg["u"]), _ptr(cfg["u_row"]),
                                                 R, S, I, _ptr(z_fine), C.c_void_p(0), _stream()), "nfl_sample_pdf")
        use_t = cfg["use_t"]
        of = _run_pass(f_f, rays, F, z=z_fine, noise=cfg["noise_f"], noise_std=cfg["noise_std"], a_emb=a_emb,
                       t_emb=t_emb if use_t else None, view_dir=cfg["view_dir"], white_back=cfg["white_back"],
                       test_extras=test_time, field_raw=raw, stash=train, pe_w_xyz=cfg["pe_w_xyz"],
                       pe_w_dir=cfg["pe_w_dir"], loss=cfg["loss"] if train else None, loss_slot=1, bprec=cfg["bprec"])
        result["weights_fine"] = of["weights"]
        result["opacity_fine"] = of["opacity"]
        if use_t:
            result["transient_sigmas"] = of["transient_sigmas"]
            result["beta"] = of["beta"]
            result["_rgb_fine_static"] = of["rgb_static"]
            result["_rgb_fine_transient"] = of["rgb_transient"]
            result["rgb_fine"] = of["rgb"]
            if test_time:
                result["rgb_fine_static"] = of["rgb_static_only"]
                result["depth_fine_static"] = of["depth_static_only"]
                result["rgb_fine_transient"] = of["rgb_transient_only"]
                result["depth_fine_transient"] = of["depth_transient_only"]
        else:
            result["rgb_fine"] = of["rgb"]
        result["depth_fine"] = of["depth"]
        if raw:
            result["_field_raw_fine"] = of["field_raw"]
            result["_z_fine"] = z_fine
        if train:
            saved["fine"] = dict(z=z_fine, field_raw=of["field_raw"], act=of["act_stash"], noise=cfg["noise_f"],
                                 use_t=use_t, n=F, seed_rgb=of.get("seed_rgb"), seed_beta=of.get("seed_beta"))
    return result, (saved if train else None)


def _g(grads, keys, name):
    """Contiguous fp32 gradient of output `name`, or None."""
    if name not in keys:
        return None
    g = grads[keys.index(name)]
    return None if g is None else g.contiguous()


def _backward_pass(field, rays, st, typ, keys, grads, cfg, want_latents, g_rays=None):
    """One pass (coarse or fine) of the hand-written backward.  Returns (flat fp32 parameter
    gradient arena views per parameter, g_a_emb, g_t_emb)."""
    L = _lib.lib()
    R, N, dev = rays.shape[0], st["n"], rays.device
    use_t = bool(st["use_t"])
    head = torch.empty(R * N, 9, dtype=torch.float32, device=dev)
    ca = _lib.CompBwdArgs()
    ca.d_field_raw, ca.d_z, ca.d_noise = _ptr(st["field_raw"]), _ptr(st["z"]), _ptr(st["noise"])
    ca.noise_std, ca.n_rays, ca.n_samples = float(cfg["noise_std"]), R, N
    ca.use_transient, ca.white_back = int(use_t), int(bool(cfg["white_back"]))
    if cfg["loss"] is not None:
        # fused NerfWLoss: the forward epilogue left d loss / d rgb (and d beta) per ray; s_l's gradient w.r.t. every
        # transient density is the constant coef * lambda_u / (R N); everything is scaled by the upstream gradient of
        # the total on the device (d_go)
        keep = [None, None, st["seed_rgb"], None, None, st["seed_beta"] if use_t else None, None, None]
        go = _g(grads, keys, "_nerfw_loss")
        if go is None:
            go = torch.zeros((), dtype=torch.float32, device=dev)
        keep.append(go)
        ca.d_go = _ptr(go)
        if use_t:
            ca.g_tsig_const = float(cfg["loss"]["coef"]) * float(cfg["loss"]["lambda_u"]) / float(R * N)
    else:
        keep = [_g(grads, keys, f"weights_{typ}"), _g(grads, keys, f"opacity_{typ}"), _g(grads, keys, f"rgb_{typ}"),
                _g(grads, keys, f"depth_{typ}")]
        if use_t:
            keep += [_g(grads, keys, "transient_sigmas"), _g(grads, keys, "beta"), _g(grads, keys, "_rgb_fine_static"),
                     _g(grads, keys, "_rgb_fine_transient")]
        else:
            keep += [None, None, None, None]
    (ca.g_weights, ca.g_opacity, ca.g_rgb, ca.g_depth, ca.g_transient_sigmas, ca.g_beta, ca.g_rgb_static,
     ca.g_rgb_transient) = [_ptr(k) for k in keep[:8]]
    ca.d_head_grads = _ptr(head)
    gmax = torch.empty(_lib.NFL_GMAX_SLOTS, dtype=torch.float32, device=dev)   # loss scale source of this pass (zeroed by the call)
    ca.d_gmax = _ptr(gmax)
    _lib.check(L.nfl_composite_backward(C.byref(ca), _stream()), "nfl_composite_backward")

    grad_stash = torch.empty(L.nfl_grad_stash_bytes(C.byref(field.desc), R, N, cfg["bprec"]), dtype=torch.uint8, device=dev)
    g_a = g_t = None
    tables = cfg["latent_tables"]
    arena = cfg["arena"]

    def latent_grad(table, n_vocab, width):
        # with a GradArena the table gradient is accumulated in place (the view is zeroed first, as a fresh tensor would be)
        v = arena.view(table) if (arena is not None and tables and table is not None) else None
        if v is not None:
            return v.zero_()
        return torch.zeros(n_vocab if tables else R, width, dtype=torch.float32, device=dev)

    if want_latents and field.desc.encode_appearance:
        g_a = latent_grad(cfg.get("table_a"), cfg.get("n_vocab_a"), field.desc.n_a)
    if want_latents and use_t:
        g_t = latent_grad(cfg.get("table_t"), cfg.get("n_vocab_t"), field.desc.n_tau)
    da = _lib.DgradArgs()
    da.d_head_grads, da.d_act_stash, da.d_grad_stash = _ptr(head), _ptr(st["act"]), _ptr(grad_stash)
    da.n_rays, da.n_samples, da.use_transient = R, N, int(use_t)
    da.d_g_a_emb, da.d_g_t_emb, da.d_gmax = _ptr(g_a), _ptr(g_t), _ptr(gmax)
    da.rounding_seed = _rounding_seed
    if tables and want_latents:
        da.d_latent_row = _ptr(cfg["ts"])       # the scatter-add into the table gradients happens in the kernel
    if g_rays is not None:
        da.d_g_rays, da.d_rays, da.d_z = _ptr(g_rays), _ptr(rays), _ptr(st["z"])
        da.dir_is_data = int(cfg["view_dir"] is not None)
        da.d_pe_w_xyz, da.d_pe_w_dir = _ptr(cfg["pe_w_xyz"]), _ptr(cfg["pe_w_dir"])
    bp = field.ensure_bwd_packed(cfg["rays_grad"], cfg["bprec"])
    _lib.check(L.nfl_mlp_dgrad(bp["h"], _ptr(bp["d"]), _ptr(bp["packed"]), C.byref(da), _stream()), "nfl_mlp_dgrad")

    plist = field.param_list()
    n_scr = L.nfl_wgrad_scratch_bytes() // 4       # composition scratch (G: include/nerf_fl_amd.h, nfl_mlp_wgrad)
    fg = _lib.FieldGrads()
    views = []
    # the call's workspace (G + the workgroups' partial sums, 68 MB): one per field, reused by every step
    if field.wg_scratch is None:
        field.wg_scratch = torch.empty(n_scr, dtype=torch.float32, device=dev)
    scratch = field.wg_scratch
    if arena is not None and all(arena.view(p) is not None for _, w, b in plist for p in (w, b)):
        # the caller's GradArena owns the gradient memory (p.grad are views of it): written in place, nothing returned
        for i, w, b in plist:
            fg.weight[i], fg.bias[i] = arena.view(w).data_ptr(), arena.view(b).data_ptr()
            views += [None, None]
    else:
        n_par = sum((w.numel() + 3) // 4 * 4 + (b.numel() + 3) // 4 * 4 for _, w, b in plist)
        block = torch.empty(n_par, dtype=torch.float32, device=dev)      # one allocation for the field's gradients (written by the call)
        off = 0
        for i, w, b in plist:
            gw = block[off:off + w.numel()].view_as(w)
            off += (w.numel() + 3) // 4 * 4
            gb = block[off:off + b.numel()].view_as(b)
            off += (b.numel() + 3) // 4 * 4
            fg.weight[i], fg.bias[i] = gw.data_ptr(), gb.data_ptr()
            views += [gw, gb]
    h_wp, d_wp = field.wgrad_plan(use_t)
    fp, _keep = field._field_params()          # the fp32 weights the forward ran with (read by the composition)
    _lib.check(L.nfl_mlp_wgrad(h_wp, _ptr(d_wp), _ptr(st["act"]), _ptr(grad_stash), _ptr(gmax), R, N, cfg["bprec"], C.byref(fp),
                               C.c_void_p(scratch.data_ptr()), C.byref(fg), _stream()),
               "nfl_mlp_wgrad")
    return views, g_a, g_t


class _RenderRaysFn(torch.autograd.Function):
    """autograd boundary: inputs are (cfg, rays, a_emb, t_emb, *parameters of coarse then fine);
    gradients are produced by the HIP backward (composite -> dgrad -> wgrad) and handed to the
    very nn.Parameter objects the optimizer / DDP hold."""

    @staticmethod
    def forward(ctx, cfg, rays, a_emb, t_emb, *params):
        ctx.set_materialize_grads(False)     # outputs the loss does not use arrive as None, not as zero tensors to fill and read
        rays = _f32c(rays, "rays")
        if cfg["latent_tables"]:          # a_emb / t_emb are the embedding tables: one gather each
            a_rows = None if a_emb is None else a_emb.detach().index_select(0, cfg["ts"])
            t_rows = None if t_emb is None else t_emb.detach().index_select(0, cfg["ts"])
            cfg["n_vocab_a"] = 0 if a_emb is None else a_emb.shape[0]
            cfg["n_vocab_t"] = 0 if t_emb is None else t_emb.shape[0]
        else:
            a_rows = None if a_emb is None else _f32c(a_emb, "a_embedded")
            t_rows = None if t_emb is None else _f32c(t_emb, "t_embedded")
        result, saved = _forward(cfg, rays, a_rows, t_rows, train=True)
        cfg["f_c"].ensure_bwd_packed(cfg["rays_grad"], cfg["bprec"])
        cfg["f_c"].wgrad_plan(False)
        if cfg["f_f"] is not None:
            cfg["f_f"].ensure_bwd_packed(cfg["rays_grad"], cfg["bprec"])
            cfg["f_f"].wgrad_plan(cfg["use_t"])
        if cfg["loss"] is not None:
            terms = cfg["loss"]["losses"]
            result["_nerfw_terms"] = terms              # (4,) = c_l, f_l, b_l, s_l (terms a configuration lacks stay 0)
            result["_nerfw_loss"] = terms.sum()         # the differentiable total
        keys = [k for k in result if not k.startswith("_field_raw") and k != "_z_fine"]
        if cfg["loss"] is not None:                     # only the total carries gradient in this mode
            ctx.mark_non_differentiable(*[result[k] for k in keys if k != "_nerfw_loss"])
        ctx.cfg, ctx.keys, ctx.saved, ctx.rays = cfg, keys, saved, rays
        # the backward reads the fp32 weights again (dgrad stream, composed gradients): they must still be the forward's
        ctx.param_keys = [f.current_key() for f in (cfg["f_c"], cfg["f_f"]) if f is not None]
        ctx.a_emb, ctx.t_emb = a_emb, t_emb
        ctx.extra = {k: v for k, v in result.items() if k not in keys}
        return tuple(result[k] for k in keys)

    @staticmethod
    def backward(ctx, *grads):
        cfg, keys, saved, rays = ctx.cfg, ctx.keys, ctx.saved, ctx.rays
        if ctx.param_keys != [f.current_key() for f in (cfg["f_c"], cfg["f_f"]) if f is not None]:
            raise RuntimeError("nerf_fl_amd.render_rays: a parameter of the field was modified in place between the forward "
                               "and the backward pass; the hand-written backward needs the weights the forward ran with")
        with torch.cuda.device(rays.device):
            out = []
            g_a = g_t = None
            g_rays = torch.zeros_like(rays) if cfg["rays_grad"] else None
            vc, _, _ = _backward_pass(cfg["f_c"], rays, saved["coarse"], "coarse", keys, grads, cfg, False, g_rays)
            out += vc
            if cfg["f_f"] is not None:
                vf, g_a, g_t = _backward_pass(cfg["f_f"], rays, saved["fine"], "fine", keys, grads, cfg, True, g_rays)
                out += vf
        ga = g_a if (ctx.a_emb is not None and ctx.needs_input_grad[2]) else None
        gt = g_t if (ctx.t_emb is not None and ctx.needs_input_grad[3]) else None
        arena = cfg["arena"]
        if arena is not None:       # gradients written in place into the arena's views: make sure p.grad IS that view
            arena.attach()
            if cfg["latent_tables"]:
                ga = None if (ga is not None and arena.view(cfg.get("table_a")) is not None) else ga
                gt = None if (gt is not None and arena.view(cfg.get("table_t")) is not None) else gt
        return (None, g_rays if ctx.needs_input_grad[1] else None, ga, gt, *out)


def render_rays(models, embeddings, rays, ts, N_samples=64, use_disp=False, perturb=0, noise_std=1,
                N_importance=0, chunk=1024 * 32, white_back=False, test_time=False, **kwargs):
    """See the reference docstring (models/rendering.py:63-81); `chunk` is accepted and
    ignored -- the fused kernel never materialises per-sample tensors, so there is
    nothing to chunk."""
    if isinstance(rays, CameraRays):          # rays generated in the kernel prologue (build-defined; inference only)
        rays_in, rays_grad = rays, False
    else:
        rays_in = rays[:, :8] if rays.shape[1] > 8 else rays
        rays_grad = bool(isinstance(rays_in, torch.Tensor) and rays_in.requires_grad and torch.is_grad_enabled())
        rays = _f32c(rays_in, "rays")
        if rays.dim() != 2 or rays.shape[1] != 8:
            raise ValueError("rays must be (N_rays, 8): origin, direction, near, far")
    R, dev = rays.shape[0], rays.device
    with torch.cuda.device(dev):
        n_xyz, n_dir = _n_freqs(embeddings["xyz"]), _n_freqs(embeddings["dir"])
        S, I = int(N_samples), int(N_importance)
        F = S + I
        test_time = bool(test_time)
        cfg = dict(S=S, I=I, use_disp=bool(use_disp), perturb=float(perturb), noise_std=float(noise_std),
                   white_back=bool(white_back), test_time=test_time, raw=bool(kwargs.get("_field_raw", False)),
                   view_dir=None, perturb_rand=None, noise_c=None, noise_f=None, u=None, u_row=None,
                   use_t=False, f_f=None, rays_grad=rays_grad, pe_w_xyz=None, pe_w_dir=None, z_fine=None, loss=None,
                   latent_tables=False, ts=None, arena=kwargs.get("grad_arena"), bprec=_BPREC[_backward])
        if kwargs.get("loss_target") is not None:
            # build-defined: NerfWLoss (losses.py:35-50) fused into the per-ray epilogue of the training passes.  The result
            # gains `_nerfw_loss` (scalar, the only output that carries gradient then) and `_nerfw_terms` (4,)
            cfg["loss"] = dict(target=_f32c(kwargs["loss_target"], "loss_target", (R, 3)),
                               losses=torch.zeros(4, dtype=torch.float32, device=dev),
                               coef=float(kwargs.get("loss_coef", 1.0)), lambda_u=float(kwargs.get("lambda_u", 0.01)))
        if getattr(models["coarse"], "refine_pose", False):
            # BARF (reference rendering.py:105-108, 235-238): coarse-to-fine weights of both encodings
            epoch = kwargs.get("current_epoch")
            if epoch is None:
                raise KeyError("current_epoch")
            cfg["pe_w_xyz"] = embeddings["xyz"].weights(epoch).to(dev) if hasattr(embeddings["xyz"], "weights") else \
                torch.tensor([float(embeddings["xyz"].barf_weight(f, epoch)) for f in embeddings["xyz"].freqs], device=dev)
            cfg["pe_w_dir"] = embeddings["dir"].weights(epoch).to(dev) if hasattr(embeddings["dir"], "weights") else \
                torch.tensor([float(embeddings["dir"].barf_weight(f, epoch)) for f in embeddings["dir"].freqs], device=dev)
        if kwargs.get("view_dir") is not None:
            if kwargs["view_dir"].requires_grad and torch.is_grad_enabled():
                raise NotImplementedError("gradient w.r.t. `view_dir` (no caller of the reference asks for it); detach it")
            # with rays.requires_grad the direction encoding is data then (rendering.py:236-238): dgrad leaves it out of rays.grad
            cfg["view_dir"] = _f32c(kwargs["view_dir"], "view_dir", (R, 3))
        cfg["f_c"] = _field(models["coarse"], n_xyz, n_dir, dev, pack=False)      # packed below, all streams in one launch
        # random draws, in the reference's order (rendering.py:258, 151, 30, 151)
        if perturb > 0:
            pr = kwargs.get("perturb_rand")
            cfg["perturb_rand"] = torch.rand(R, S, device=dev) if pr is None else _f32c(pr, "perturb_rand", (R, S))
        nc = kwargs.get("noise_coarse")
        nc = torch.randn(R, S, device=dev) if nc is None else _f32c(nc, "noise_coarse", (R, S))
        cfg["noise_c"] = nc if noise_std != 0 else None
        a_emb = t_emb = None
        params = [p for _, w, b in cfg["f_c"].param_list() for p in (w, b)]
        if I > 0:
            if S < 3:
                raise ValueError("N_samples must be >= 3 when N_importance > 0")
            if kwargs.get("z_fine") is not None:
                cfg["z_fine"] = _f32c(kwargs["z_fine"], "z_fine", (R, F))
            if perturb == 0:
                cfg["u_row"] = _linspace(I, dev)
            else:
                u = kwargs.get("u")
                cfg["u"] = torch.rand(R, I, device=dev) if u is None else _f32c(u, "u", (R, I))
            fine = models["fine"]
            cfg["f_f"] = f_f = _field(fine, n_xyz, n_dir, dev, pack=False)
            params += [p for _, w, b in f_f.param_list() for p in (w, b)]
            cfg["use_t"] = bool(kwargs.get("output_transient", True) and fine.encode_transient)
            # Latent codes (rendering.py:276-286).  When every code in use comes from a plain nn.Embedding and a gradient
            # is wanted, the autograd.Function takes the TABLES: the lookup is one gather and the backward's scatter-add
            # into the (N_vocab, n) table gradient happens inside the dgrad kernel (nfl_dgrad_args::d_latent_row) instead
            # of torch's embedding_backward (2 x 59 us at 1024 rays, 6 % of that step).
            need_a, need_t = bool(fine.encode_appearance), cfg["use_t"]
            from_table = {"a": need_a and "a_embedded" not in kwargs, "t": need_t and "t_embedded" not in kwargs}
            cfg["latent_tables"] = bool(
                torch.is_grad_enabled() and not test_time and (need_a or need_t)
                and all(from_table[k] and _plain_embedding(embeddings[k]) and embeddings[k].weight.requires_grad
                        for k, need in (("a", need_a), ("t", need_t)) if need))
            if cfg["latent_tables"]:
                cfg["ts"] = ts.detach().to(device=dev, dtype=torch.int64).contiguous()
                if cfg["ts"].shape != (R,):
                    raise ValueError(f"ts must be ({R},)")
                a_emb = cfg["table_a"] = embeddings["a"].weight if need_a else None
                t_emb = cfg["table_t"] = embeddings["t"].weight if need_t else None
            else:
                if need_a:
                    a_emb = kwargs["a_embedded"] if "a_embedded" in kwargs else embeddings["a"](ts)
                    if tuple(a_emb.shape) != (R, f_f.desc.n_a):
                        raise ValueError(f"a_embedded must be ({R}, {f_f.desc.n_a})")
                if need_t:
                    t_emb = kwargs["t_embedded"] if "t_embedded" in kwargs else embeddings["t"](ts)
                    if tuple(t_emb.shape) != (R, f_f.desc.n_tau):
                        raise ValueError(f"t_embedded must be ({R}, {f_f.desc.n_tau})")
            if not cfg["use_t"]:            # the transient branch draws no density noise (rendering.py:146-149)
                nf = kwargs.get("noise_fine")
                nf = torch.randn(R, F, device=dev) if nf is None else _f32c(nf, "noise_fine", (R, F))
                cfg["noise_f"] = nf if noise_std != 0 else None

        if R == 0:          # nothing to launch: the reference returns empty tensors of the right shapes
            shp = lambda k: ((0, S if k.endswith("coarse") else F) if k.startswith(("weights", "transient_sigmas"))
                             else (0, 3) if "rgb" in k else (0,))
            return {k: torch.empty(shp(k), dtype=torch.float32, device=dev) for k in _result_keys(cfg)}
        needs_grad = torch.is_grad_enabled() and (
            rays_grad or any(p.requires_grad for p in params)
            or any(t is not None and t.requires_grad for t in (a_emb, t_emb)))
        _pack_streams([cfg["f_c"], cfg["f_f"]], bwd=needs_grad and not test_time and _precision == "f16x3", rays_grad=rays_grad)
        if cfg["loss"] is not None and not (torch.is_grad_enabled() and not test_time):
            raise RuntimeError("loss_target fuses the loss into the TRAINING passes: call it with gradients enabled")
        if needs_grad:
            if isinstance(rays, CameraRays):
                raise RuntimeError("CameraRays is an inference input; call render_rays under torch.no_grad()")
            if test_time:
                raise RuntimeError("test_time=True is an inference mode; call it under torch.no_grad()")
            if _precision != "f16x3":
                raise RuntimeError("training needs the accurate mode: nerf_fl_amd.set_precision('f16x3') "
                                   "(the fast 'f16' mode is inference-only)")
            outs = _RenderRaysFn.apply(cfg, rays_in if rays_grad else rays, a_emb, t_emb, *params)
            keys = [k for k in _result_keys(cfg)] + (["_nerfw_terms", "_nerfw_loss"] if cfg["loss"] is not None else [])
            if kwargs.get("check_finite"):
                check_status(dev)
            return dict(zip(keys, outs))
        a_c = None if a_emb is None else _f32c(a_emb, "a_embedded")
        t_c = None if t_emb is None else _f32c(t_emb, "t_embedded")
        result, _ = _forward(cfg, rays, a_c, t_c, train=False)
        if kwargs.get("check_finite"):       # build-defined kwarg: synchronises; see check_status()
            check_status(dev)
    return result


def _result_keys(cfg):
    keys = ["weights_coarse", "opacity_coarse"]
    if not cfg["test_time"]:
        keys += ["rgb_coarse", "depth_coarse"]
    if cfg["I"] > 0:
        keys += ["weights_fine", "opacity_fine"]
        if cfg["use_t"]:
            keys += ["transient_sigmas", "beta", "_rgb_fine_static", "_rgb_fine_transient", "rgb_fine"]
            if cfg["test_time"]:
                keys += ["rgb_fine_static", "depth_fine_static", "rgb_fine_transient", "depth_fine_transient"]
        else:
            keys += ["rgb_fine"]
        keys += ["depth_fine"]
    return keys
