"""CPU oracle for the NeRF-W ray renderer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`nerf_fl_amd/`) may
import this module; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` do, and only as the checker / timed baseline.

This is a restatement (plain eager PyTorch, fp32, CPU) of the algorithm of the
reference's hot path, written from SURVEY.md section 8 and from reading
  models/nerf.py:6-32      (PosEmbedding)           -> posenc()
  models/nerf.py:81-212    (NeRF.__init__/forward)  -> field_forward()
  models/rendering.py:7-46 (sample_pdf)             -> sample_pdf()
  models/rendering.py:83-226 (inference)            -> composite()
  models/rendering.py:49-289 (render_rays)          -> render_rays()
  losses.py:18-50          (NerfWLoss)              -> nerfw_loss()
It is *pinned*: `tests/golden/make_golden.py` imports the real reference in the
build container and stores its inputs/outputs in `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks this module against every one of them.

Conventions
-----------
* A field (one NeRF MLP) is a `FieldSpec` + a dict of fp32 tensors keyed by the
  reference's state_dict names ("xyz_encoding_1.0.weight", ...).
* All randomness is injected: `perturb_rand` (R,S) ~ U[0,1), `noise_coarse`
  (R,S) ~ N(0,1), `u` (R,I) ~ U[0,1), `noise_fine` (R,F) ~ N(0,1).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

Tensor = torch.Tensor

TRUNK_DEPTH = 8
TRUNK_WIDTH = 256
SKIP_AT = 4  # zero-based trunk layer that re-reads the encoded position


@dataclass(frozen=True)
class FieldSpec:
    """Static description of one field (reference: models/nerf.py:81-120)."""

    typ: str = "coarse"            # 'coarse' | 'fine'
    n_emb_xyz: int = 10
    n_emb_dir: int = 4
    encode_appearance: bool = False
    n_a: int = 48
    encode_transient: bool = False
    n_tau: int = 16
    beta_min: float = 0.03

    def __post_init__(self):
        # the coarse field never sees latent codes (models/nerf.py:115,117)
        if self.typ == "coarse":
            object.__setattr__(self, "encode_appearance", False)
            object.__setattr__(self, "encode_transient", False)

    @property
    def c_xyz(self) -> int:
        return 6 * self.n_emb_xyz + 3

    @property
    def c_dir(self) -> int:
        return 6 * self.n_emb_dir + 3

    @property
    def c_a(self) -> int:
        return self.n_a if self.encode_appearance else 0


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------
def field_param_shapes(spec: FieldSpec) -> Dict[str, tuple]:
    """Name -> shape for every parameter, in the reference's registration order
    (models/nerf.py:121-151)."""
    W, h = TRUNK_WIDTH, TRUNK_WIDTH // 2
    shp: Dict[str, tuple] = {}
    for i in range(TRUNK_DEPTH):
        fan_in = spec.c_xyz if i == 0 else (W + spec.c_xyz if i == SKIP_AT else W)
        shp[f"xyz_encoding_{i + 1}.0.weight"] = (W, fan_in)
        shp[f"xyz_encoding_{i + 1}.0.bias"] = (W,)
    shp["xyz_encoding_final.weight"] = (W, W)
    shp["xyz_encoding_final.bias"] = (W,)
    shp["dir_encoding.0.weight"] = (h, W + spec.c_dir + spec.c_a)
    shp["dir_encoding.0.bias"] = (h,)
    shp["static_sigma.0.weight"] = (1, W)
    shp["static_sigma.0.bias"] = (1,)
    shp["static_rgb.0.weight"] = (3, h)
    shp["static_rgb.0.bias"] = (3,)
    if spec.encode_transient:
        shp["transient_encoding.0.weight"] = (h, W + spec.n_tau)
        shp["transient_encoding.0.bias"] = (h,)
        for j in (2, 4, 6):
            shp[f"transient_encoding.{j}.weight"] = (h, h)
            shp[f"transient_encoding.{j}.bias"] = (h,)
        shp["transient_sigma.0.weight"] = (1, h)
        shp["transient_sigma.0.bias"] = (1,)
        shp["transient_rgb.0.weight"] = (3, h)
        shp["transient_rgb.0.bias"] = (3,)
        shp["transient_beta.0.weight"] = (1, h)
        shp["transient_beta.0.bias"] = (1,)
    return shp


def make_field_params(spec: FieldSpec, seed: int, regime: str = "default") -> Dict[str, Tensor]:
    """Build-owned, platform-independent seeded initialiser (numpy PCG64).

    regime 'default'  : U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases.
    regime 'sharp'    : same, then the sigma heads are scaled (weight x30, bias -3)
                        so densities are peaky and importance sampling is exercised.
    """
    rng = np.random.default_rng(seed)
    out: Dict[str, Tensor] = {}
    for name, shape in field_param_shapes(spec).items():
        wname = name.rsplit(".", 1)[0] + ".weight"
        fan_in = field_param_shapes(spec)[wname][1]
        bound = 1.0 / np.sqrt(fan_in)
        arr = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        out[name] = torch.from_numpy(arr)
    if regime == "sharp":
        for head in ("static_sigma", "transient_sigma"):
            if f"{head}.0.weight" in out:
                out[f"{head}.0.weight"] = out[f"{head}.0.weight"] * 30.0
                out[f"{head}.0.bias"] = out[f"{head}.0.bias"] - 3.0
    elif regime != "default":
        raise ValueError(regime)
    return out


def make_embedding_table(n_vocab: int, dim: int, seed: int) -> Tensor:
    """N(0,1) latent table (torch.nn.Embedding's default init), numpy-seeded."""
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal((n_vocab, dim)).astype(np.float32))


# --------------------------------------------------------------------------
# A1: positional encoding          (reference models/nerf.py:6-32)
# --------------------------------------------------------------------------
def posenc(x: Tensor, n_freqs: int, weights: Optional[Tensor] = None) -> Tensor:
    """[x | sin(2^0 x) | cos(2^0 x) | ... ]: x first, sin before cos,
    frequency-major, 3 columns per block.  `weights` (n_freqs,) = BARF coarse-to-fine
    weights of the sin/cos blocks (reference models/nerf.py:61-77)."""
    cols = [x]
    for k in range(n_freqs):
        arg = x * float(2 ** k)          # power-of-two scaling: exact in fp32
        w = 1.0 if weights is None else float(weights[k])
        cols.append(w * arg.sin() if weights is not None else arg.sin())
        cols.append(w * arg.cos() if weights is not None else arg.cos())
    return torch.cat(cols, dim=-1)


def barf_weights(n_freqs: int, epoch, epoch_start: int = 4, epoch_end: int = 8) -> Tensor:
    """Per-frequency weights of the reference's BarfPosEmbedding (models/nerf.py:47-59, constants
    train.py:43-44), restated literally: alpha is compared with the frequency VALUE 2^k."""
    if epoch_start < epoch <= epoch_end:
        alpha = n_freqs / epoch
    elif epoch > epoch_end:
        alpha = float(n_freqs)
    else:
        alpha = 0.0
    out = []
    for k in range(n_freqs):
        freq = float(2 ** k)
        if alpha < freq:
            out.append(0.0)
        elif alpha - freq < 1:
            out.append(float((1 - torch.cos(torch.tensor((alpha - freq) * np.pi, dtype=torch.float32))) / 2))
        else:
            out.append(1.0)
    return torch.tensor(out, dtype=torch.float32)


# --------------------------------------------------------------------------
# A3: the field MLP               (reference models/nerf.py:153-212)
# --------------------------------------------------------------------------
def _lin(P: Dict[str, Tensor], name: str, x: Tensor) -> Tensor:
    return torch.nn.functional.linear(x, P[name + ".weight"], P[name + ".bias"])      # what nn.Linear calls


def _lin_relu(P: Dict[str, Tensor], name: str, x: Tensor) -> Tensor:
    # nn.Sequential(nn.Linear, nn.ReLU(True)) (models/nerf.py:124-151): the activation overwrites the layer's output
    return torch.relu_(_lin(P, name, x))


def field_forward(spec: FieldSpec, P: Dict[str, Tensor], enc_xyz: Tensor,
                  dir_a: Optional[Tensor] = None, tau: Optional[Tensor] = None,
                  sigma_only: bool = False) -> Dict[str, Tensor]:
    """Evaluate one field on already-encoded inputs.

    enc_xyz (B, c_xyz); dir_a (B, c_dir [+ n_a]); tau (B, n_tau) or None.
    Returns dict with 'sigma' (B,), and unless sigma_only 'rgb' (B,3); with tau
    also 'sigma_t' (B,), 'rgb_t' (B,3), 'beta' (B,).
    """
    h = enc_xyz
    for i in range(TRUNK_DEPTH):
        if i == SKIP_AT:
            h = torch.cat([enc_xyz, h], dim=1)       # encoded position goes FIRST
        h = _lin_relu(P, f"xyz_encoding_{i + 1}.0", h)
    out = {"sigma": torch.nn.functional.softplus(_lin(P, "static_sigma.0", h))[:, 0]}
    if sigma_only:
        return out
    feat = _lin(P, "xyz_encoding_final", h)          # no activation
    d = _lin_relu(P, "dir_encoding.0", torch.cat([feat, dir_a], dim=1))
    out["rgb"] = torch.sigmoid(_lin(P, "static_rgb.0", d))
    if tau is None:
        return out
    g = torch.cat([feat, tau], dim=1)
    for j in (0, 2, 4, 6):
        g = _lin_relu(P, f"transient_encoding.{j}", g)
    out["sigma_t"] = torch.nn.functional.softplus(_lin(P, "transient_sigma.0", g))[:, 0]
    out["rgb_t"] = torch.sigmoid(_lin(P, "transient_rgb.0", g))
    out["beta"] = torch.nn.functional.softplus(_lin(P, "transient_beta.0", g))[:, 0]
    return out


def field_forward_packed(spec: FieldSpec, P: Dict[str, Tensor], x: Tensor,
                         sigma_only: bool = False, output_transient: bool = True) -> Tensor:
    """Same column conventions as the reference module's forward():
    input [xyz | dir (+a) | tau], output [rgb, sigma (, rgb_t, sigma_t, beta)]."""
    cx, cda = spec.c_xyz, spec.c_dir + spec.c_a
    if sigma_only:
        return field_forward(spec, P, x, sigma_only=True)["sigma"][:, None]
    tau = x[:, cx + cda: cx + cda + spec.n_tau] if output_transient else None
    o = field_forward(spec, P, x[:, :cx], x[:, cx: cx + cda], tau)
    cols = [o["rgb"], o["sigma"][:, None]]
    if tau is not None:
        cols += [o["rgb_t"], o["sigma_t"][:, None], o["beta"][:, None]]
    return torch.cat(cols, dim=1)


# --------------------------------------------------------------------------
# A7: inverse-CDF importance sampling   (reference models/rendering.py:7-46)
# --------------------------------------------------------------------------
def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor, eps: float = 1e-5) -> Tensor:
    """bins (R, M+1), weights (R, M), u (R, I) in [0,1] -> samples (R, I)."""
    M = weights.shape[1]
    w = weights + eps
    pdf = w / w.sum(dim=1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), pdf.cumsum(dim=1)], dim=1)   # (R, M+1)
    hi = torch.searchsorted(cdf, u.contiguous(), right=True)                    # #{cdf <= u}
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=M)
    c0, c1 = cdf.gather(1, lo), cdf.gather(1, hi)
    b0, b1 = bins.gather(1, lo), bins.gather(1, hi)
    span = c1 - c0
    span = torch.where(span < eps, torch.ones_like(span), span)
    return b0 + (u - c0) / span * (b1 - b0)


# --------------------------------------------------------------------------
# A4: coarse depths              (reference models/rendering.py:243-259)
# --------------------------------------------------------------------------
def coarse_depths(near: Tensor, far: Tensor, n_samples: int, use_disp: bool,
                  perturb: float, perturb_rand: Optional[Tensor]) -> Tensor:
    s = torch.linspace(0, 1, n_samples)
    if use_disp:
        z = 1 / (1 / near * (1 - s) + 1 / far * s)
    else:
        z = near * (1 - s) + far * s
    z = z.expand(near.shape[0], n_samples)
    if perturb > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper = torch.cat([mid, z[:, -1:]], dim=1)
        lower = torch.cat([z[:, :1], mid], dim=1)
        z = lower + (upper - lower) * (perturb * perturb_rand)
    return z


# --------------------------------------------------------------------------
# A6: volume-rendering compositing   (reference models/rendering.py:141-226)
# --------------------------------------------------------------------------
def _transmittance(alpha: Tensor) -> Tensor:
    # T_i = prod_{j<i} (1 - alpha_j); no epsilon inside the product
    shifted = torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha], dim=1)
    return torch.cumprod(shifted[:, :-1], dim=1)


def composite(typ: str, z: Tensor, f: Dict[str, Tensor], *, noise: Optional[Tensor],
              noise_std: float, white_back: bool, test_time: bool,
              beta_min: float) -> Dict[str, Tensor]:
    """z (R,N); f holds per-sample field outputs reshaped to (R,N[,3])."""
    res: Dict[str, Tensor] = {}
    delta = torch.cat([z[:, 1:] - z[:, :-1], 1e2 * torch.ones_like(z[:, :1])], dim=1)
    transient = "sigma_t" in f
    sig = f["sigma"]
    if transient:
        a_s = 1 - torch.exp(-delta * sig)
        a_t = 1 - torch.exp(-delta * f["sigma_t"])
        alpha = 1 - torch.exp(-delta * (sig + f["sigma_t"]))
    else:
        alpha = 1 - torch.exp(-delta * torch.relu(sig + noise * noise_std))
    T = _transmittance(alpha)
    w = alpha * T
    wsum = w.sum(dim=1)
    res[f"weights_{typ}"] = w
    res[f"opacity_{typ}"] = wsum
    if transient:
        res["transient_sigmas"] = f["sigma_t"]
    if test_time and typ == "coarse":
        return res
    if transient:
        w_s, w_t = a_s * T, a_t * T
        rgb_s = (w_s[..., None] * f["rgb"]).sum(dim=1)
        if white_back:
            rgb_s = rgb_s + (1 - wsum[:, None])
        rgb_t = (w_t[..., None] * f["rgb_t"]).sum(dim=1)
        res["beta"] = (w_t * f["beta"]).sum(dim=1) + beta_min
        res["_rgb_fine_static"] = rgb_s
        res["_rgb_fine_transient"] = rgb_t
        res["rgb_fine"] = rgb_s + rgb_t
        if test_time:
            ws1 = a_s * _transmittance(a_s)
            only_s = (ws1[..., None] * f["rgb"]).sum(dim=1)
            if white_back:
                only_s = only_s + (1 - wsum[:, None])       # combined sum, on purpose
            res["rgb_fine_static"] = only_s
            res["depth_fine_static"] = (ws1 * z).sum(dim=1)
            wt1 = a_t * _transmittance(a_t)
            res["rgb_fine_transient"] = (wt1[..., None] * f["rgb_t"]).sum(dim=1)
            res["depth_fine_transient"] = (wt1 * z).sum(dim=1)
    else:
        rgb = (w[..., None] * f["rgb"]).sum(dim=1)
        if white_back:
            rgb = rgb + (1 - wsum[:, None])
        res[f"rgb_{typ}"] = rgb
    res[f"depth_{typ}"] = (w * z).sum(dim=1)
    return res


# --------------------------------------------------------------------------
# render_rays                      (reference models/rendering.py:49-289)
# --------------------------------------------------------------------------
POINT_CHUNK = 1024 * 32      # the reference's `chunk` default (rendering.py:58): points per field evaluation


def _eval_field(spec, P, xyz, dir_enc, a_emb, t_emb, sigma_only, pe_w_xyz=None, chunk=POINT_CHUNK):
    """The reference's point-chunk loop (rendering.py:98-139): the (R*N, 3) points are encoded and pushed through the
    field `chunk` rows at a time, the per-ray side inputs repeated per sample, the chunk outputs concatenated."""
    R, N = xyz.shape[:2]
    pts = xyz.reshape(-1, 3)
    B = pts.shape[0]
    dir_a = tau = None
    if not sigma_only:
        side = [dir_enc]
        if spec.encode_appearance:
            side.append(a_emb)
        dir_a = torch.cat(side, dim=1).repeat_interleave(N, dim=0)
        tau = t_emb.repeat_interleave(N, dim=0) if t_emb is not None else None
    parts = []
    for i in range(0, max(B, 1), chunk):
        enc = posenc(pts[i:i + chunk], spec.n_emb_xyz, pe_w_xyz)
        if sigma_only:
            parts.append(field_forward(spec, P, enc, sigma_only=True))
        else:
            parts.append(field_forward(spec, P, enc, dir_a[i:i + chunk], None if tau is None else tau[i:i + chunk]))
    o = parts[0] if len(parts) == 1 else {k: torch.cat([p[k] for p in parts], dim=0) for k in parts[0]}
    return {k: v.reshape(R, N, *v.shape[1:]) for k, v in o.items()}


def render_rays(spec_c: FieldSpec, P_c: Dict[str, Tensor],
                spec_f: Optional[FieldSpec], P_f: Optional[Dict[str, Tensor]],
                rays: Tensor, *, n_samples: int = 64, use_disp: bool = False,
                perturb: float = 0.0, noise_std: float = 1.0, n_importance: int = 0,
                white_back: bool = False, test_time: bool = False,
                a_emb: Optional[Tensor] = None, t_emb: Optional[Tensor] = None,
                output_transient: bool = True, view_dir: Optional[Tensor] = None,
                perturb_rand: Optional[Tensor] = None, noise_coarse: Optional[Tensor] = None,
                u: Optional[Tensor] = None, noise_fine: Optional[Tensor] = None,
                return_z: bool = False, pe_w_xyz: Optional[Tensor] = None,
                pe_w_dir: Optional[Tensor] = None, z_fine: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """z_fine (R, n_samples + n_importance): use these sorted fine depths instead of sampling them (the depths a
    reference run used, stored in the fixtures: takes the discontinuous sampler out of a comparison)."""
    R = rays.shape[0]
    o, d = rays[:, 0:3], rays[:, 3:6]
    near, far = rays[:, 6:7], rays[:, 7:8]
    dir_enc = posenc(d if view_dir is None else view_dir, spec_c.n_emb_dir, pe_w_dir)
    zeros = lambda n: torch.zeros(R, n)

    z = coarse_depths(near, far, n_samples, use_disp, perturb, perturb_rand)
    xyz = o[:, None, :] + d[:, None, :] * z[..., None]
    fc = _eval_field(spec_c, P_c, xyz, dir_enc, None, None, sigma_only=test_time, pe_w_xyz=pe_w_xyz)
    res = composite("coarse", z, fc, noise=noise_coarse if noise_coarse is not None else zeros(n_samples),
                    noise_std=noise_std, white_back=white_back, test_time=test_time,
                    beta_min=spec_c.beta_min)
    if n_importance > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        if perturb == 0:
            u = torch.linspace(0, 1, n_importance).expand(R, n_importance)
        if z_fine is None:
            zs = sample_pdf(mid, res["weights_coarse"][:, 1:-1].detach(), u)
            z = torch.sort(torch.cat([z, zs], dim=1), dim=1)[0]
        else:
            z = z_fine
        xyz = o[:, None, :] + d[:, None, :] * z[..., None]
        use_t = bool(output_transient and spec_f.encode_transient)
        ff = _eval_field(spec_f, P_f, xyz, dir_enc, a_emb, t_emb if use_t else None, sigma_only=False,
                         pe_w_xyz=pe_w_xyz)
        n_f = n_samples + n_importance
        res.update(composite("fine", z, ff, noise=noise_fine if noise_fine is not None else zeros(n_f),
                             noise_std=noise_std, white_back=white_back, test_time=test_time,
                             beta_min=spec_f.beta_min))
        if return_z:
            res["_z_fine"] = z
    return res


# --------------------------------------------------------------------------
# A9: the consumer that defines which outputs carry gradient (losses.py:18-50)
# --------------------------------------------------------------------------
def nerfw_loss(res: Dict[str, Tensor], target: Tensor, coef: float = 1.0,
               lambda_u: float = 0.01) -> Dict[str, Tensor]:
    out = {"c_l": 0.5 * ((res["rgb_coarse"] - target) ** 2).mean()}
    if "rgb_fine" in res:
        if "beta" not in res:
            out["f_l"] = 0.5 * ((res["rgb_fine"] - target) ** 2).mean()
        else:
            b = res["beta"]
            out["f_l"] = ((res["rgb_fine"] - target) ** 2 / (2 * b[:, None] ** 2)).mean()
            out["b_l"] = 3 + torch.log(b).mean()
            out["s_l"] = lambda_u * res["transient_sigmas"].mean()
    return {k: coef * v for k, v in out.items()}


def psnr(img: Tensor, ref: Tensor) -> float:
    """-10 log10(mse)   (reference metrics.py:12-13)."""
    return float(-10.0 * torch.log10(((img - ref) ** 2).mean()))


# --------------------------------------------------------------------------
# synthetic inputs shared by tests, smoke() and bench.py
# --------------------------------------------------------------------------
def make_rays(n_rays: int, seed: int, near: float = 2.0, far: float = 6.0) -> Tensor:
    """Blender-like rays: origins near (0,0,4), unit directions towards the origin."""
    rng = np.random.default_rng(seed)
    o = np.array([0.0, 0.0, 4.0]) + 0.1 * rng.standard_normal((n_rays, 3))
    tgt = 0.8 * rng.uniform(-1, 1, size=(n_rays, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    nf = np.stack([np.full(n_rays, near), np.full(n_rays, far)], axis=1)
    return torch.from_numpy(np.concatenate([o, d, nf], axis=1).astype(np.float32))


def make_rays_photo(n_rays: int, seed: int) -> Tensor:
    """Phototourism-like rays (reference datasets/phototourism.py:130-140, 173-183): cameras scattered in front of the
    scene, unit directions, and near/far DIFFERENT ON EVERY RAY (per-image depth percentiles, rescaled so that the
    largest far bound is 5)."""
    rng = np.random.default_rng(seed)
    o = np.array([0.0, 0.0, 2.5]) + rng.uniform(-1.0, 1.0, size=(n_rays, 3)) * np.array([1.5, 0.5, 0.7])
    tgt = rng.uniform(-1, 1, size=(n_rays, 3)) * np.array([1.2, 0.8, 0.5])
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    near = rng.uniform(0.05, 1.5, size=n_rays)
    far = near + rng.uniform(1.0, 3.5, size=n_rays)
    far *= 5.0 / far.max()
    near = np.minimum(near, 0.6 * far)
    return torch.from_numpy(np.concatenate([o, d, near[:, None], far[:, None]], axis=1).astype(np.float32))
