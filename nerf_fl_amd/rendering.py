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
build-defined kwargs `perturb_rand`, `noise_coarse`, `u`, `noise_fine`.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib

__all__ = ["render_rays", "set_precision", "get_precision"]

_PREC = {"f16x3": _lib.NFL_PREC_F16X3, "f16": _lib.NFL_PREC_F16}
_precision = os.environ.get("NERF_FL_AMD_PREC", "f16x3")


def set_precision(name):
    """'f16x3' (default; fp16 MFMA with split operands, fp32-class accuracy) or 'f16' (fast)."""
    global _precision
    if name not in _PREC:
        raise ValueError(f"precision must be one of {sorted(_PREC)}")
    _precision = name


def get_precision():
    return _precision


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t, name, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"nerf_fl_amd.render_rays: `{name}` must be a CUDA/HIP tensor (this build has no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"`{name}` must be float32, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"`{name}` has shape {tuple(t.shape)}, expected {tuple(shape)}")
    t = t.detach()
    if not t.is_contiguous() or t.data_ptr() % 16:
        t = t.contiguous().clone() if t.data_ptr() % 16 else t.contiguous()
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
        if getattr(model, "refine_pose", False):
            raise NotImplementedError("BARF-weighted positional encoding (--refine_pose) is not built yet")
        nbytes = L.nfl_plan_bytes(C.byref(self.desc))
        self.h_plan = C.create_string_buffer(nbytes)
        _lib.check(L.nfl_plan_build(C.byref(self.desc), prec, self.h_plan, nbytes), "nfl_plan_build")
        self.d_plan = torch.frombuffer(bytearray(self.h_plan.raw), dtype=torch.uint8).to(device)
        self.packed_bytes = L.nfl_packed_bytes(C.byref(self.desc), prec)
        self.packed = torch.empty(self.packed_bytes, dtype=torch.uint8, device=device)
        self.key = None

    def ensure_packed(self):
        model = self.model_ref()
        params = dict(model.named_parameters())
        key = tuple((n, p.data_ptr(), p._version) for n, p in params.items())
        if key == self.key:
            return
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
        _lib.check(_lib.lib().nfl_pack_field(self.h_plan, _ptr(self.d_plan), C.byref(fp), _ptr(self.packed),
                                             self.packed_bytes, _stream()), "nfl_pack_field")
        self.key = key


_fields = weakref.WeakKeyDictionary()
_lin_cache = {}


def _field(model, n_emb_xyz, n_emb_dir, device):
    prec = _PREC[_precision]
    slot = _fields.setdefault(model, {})
    k = (prec, str(device), n_emb_xyz, n_emb_dir)
    if k not in slot:
        slot[k] = _PackedField(model, n_emb_xyz, n_emb_dir, prec, device)
    f = slot[k]
    f.ensure_packed()
    return f


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
              white_back=False, test_extras=False, want_rgb=True, want_z=False, field_raw=False):
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
    if field_raw:
        out["field_raw"] = torch.zeros(R * n_samples, 9, dtype=torch.float32, device=dev)
    a = _lib.PassArgs()
    a.d_rays, a.d_view_dir = _ptr(rays), _ptr(view_dir)
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
    _lib.check(_lib.lib().nfl_render_pass(field.h_plan, _ptr(field.d_plan), _ptr(field.packed), C.byref(a), _stream()),
               "nfl_render_pass")
    return out


def render_rays(models, embeddings, rays, ts, N_samples=64, use_disp=False, perturb=0, noise_std=1,
                N_importance=0, chunk=1024 * 32, white_back=False, test_time=False, **kwargs):
    """See the reference docstring (models/rendering.py:63-81); `chunk` is accepted and
    ignored -- the fused kernel never materialises per-sample tensors, so there is
    nothing to chunk."""
    if torch.is_grad_enabled() and any(p.requires_grad for m in models.values() for p in m.parameters()):
        raise NotImplementedError(
            "nerf_fl_amd.render_rays: the hand-written backward is not built yet; call under torch.no_grad()")
    rays = _f32c(rays[:, :8] if rays.shape[1] > 8 else rays, "rays")
    if rays.dim() != 2 or rays.shape[1] != 8:
        raise ValueError("rays must be (N_rays, 8): origin, direction, near, far")
    R, dev = rays.shape[0], rays.device
    with torch.cuda.device(dev):
        n_xyz, n_dir = _n_freqs(embeddings["xyz"]), _n_freqs(embeddings["dir"])
        coarse = models["coarse"]
        view_dir = kwargs.get("view_dir")
        if view_dir is not None:
            view_dir = _f32c(view_dir, "view_dir", (R, 3))
        S = int(N_samples)
        raw = bool(kwargs.get("_field_raw", False))
        result = {}

        # ---- coarse pass (reference rendering.py:243-265)
        f_c = _field(coarse, n_xyz, n_dir, dev)
        perturb_rand = None
        if perturb > 0:
            perturb_rand = kwargs.get("perturb_rand")
            perturb_rand = torch.rand(R, S, device=dev) if perturb_rand is None else _f32c(perturb_rand, "perturb_rand", (R, S))
        noise_c = kwargs.get("noise_coarse")
        noise_c = torch.randn(R, S, device=dev) if noise_c is None else _f32c(noise_c, "noise_coarse", (R, S))
        oc = _run_pass(f_c, rays, S, lin=_linspace(S, dev), perturb_rand=perturb_rand, perturb=perturb,
                       use_disp=use_disp, noise=noise_c if noise_std != 0 else None, noise_std=noise_std,
                       view_dir=view_dir, sigma_only=bool(test_time), white_back=white_back,
                       want_z=N_importance > 0, field_raw=raw)
        result["weights_coarse"] = oc["weights"]
        result["opacity_coarse"] = oc["opacity"]
        if not test_time:
            result["rgb_coarse"] = oc["rgb"]
            result["depth_coarse"] = oc["depth"]
        if raw:
            result["_field_raw_coarse"] = oc["field_raw"]

        if N_importance > 0:
            # ---- hierarchical sampling (reference rendering.py:267-273, 7-46)
            I = int(N_importance)
            F = S + I
            if S < 3:
                raise ValueError("N_samples must be >= 3 when N_importance > 0")
            u = u_row = None
            if perturb == 0:
                u_row = _linspace(I, dev)
            else:
                u = kwargs.get("u")
                u = torch.rand(R, I, device=dev) if u is None else _f32c(u, "u", (R, I))
            z_fine = torch.empty(R, F, dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().nfl_sample_pdf(_ptr(oc["z"]), _ptr(oc["weights"]), _ptr(u), _ptr(u_row), R, S, I,
                                                 _ptr(z_fine), C.c_void_p(0), _stream()), "nfl_sample_pdf")
            # ---- fine pass (reference rendering.py:275-287)
            fine = models["fine"]
            f_f = _field(fine, n_xyz, n_dir, dev)
            a_emb = t_emb = None
            if fine.encode_appearance:
                a_emb = kwargs["a_embedded"] if "a_embedded" in kwargs else embeddings["a"](ts)
                a_emb = _f32c(a_emb, "a_embedded", (R, f_f.desc.n_a))
            use_t = bool(kwargs.get("output_transient", True) and fine.encode_transient)
            if use_t:
                t_emb = kwargs["t_embedded"] if "t_embedded" in kwargs else embeddings["t"](ts)
                t_emb = _f32c(t_emb, "t_embedded", (R, f_f.desc.n_tau))
            noise_f = None
            if not use_t:
                noise_f = kwargs.get("noise_fine")
                noise_f = torch.randn(R, F, device=dev) if noise_f is None else _f32c(noise_f, "noise_fine", (R, F))
            of = _run_pass(f_f, rays, F, z=z_fine, noise=noise_f if noise_std != 0 else None, noise_std=noise_std,
                           a_emb=a_emb, t_emb=t_emb, view_dir=view_dir, white_back=white_back,
                           test_extras=bool(test_time), field_raw=raw)
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
    return result
