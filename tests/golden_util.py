"""Helpers shared by the CPU (oracle) and GPU (parity) tests: load a golden
fixture written by tests/golden/make_golden.py and rebuild the inputs it was
generated from (weights come from the seeded build-owned initialiser)."""
import glob
import json
import os

import numpy as np
import torch

from oracle import nerfw_oracle as orc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAINED = os.path.join(GOLDEN_DIR, "w_trained.npz")     # weights of the reference after train_reference() (make_golden.py)


def trained_params(spec):
    z = np.load(TRAINED, allow_pickle=False)
    tag = "fine" if (spec.encode_appearance or spec.encode_transient) else "coarse"
    return {k[len(tag) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + ".")}


def field_params(spec, seed, regime):
    return trained_params(spec) if regime == "trained" else orc.make_field_params(spec, seed, regime)


def embedding_table(cfg, which):
    """The (N_vocab, dim) latent table a fixture was generated with; which = 'a' | 't'."""
    if cfg["regime"] == "trained":
        return torch.from_numpy(np.load(TRAINED, allow_pickle=False)["table_" + which])
    dim, off = (cfg.get("n_a", 48), 4) if which == "a" else (cfg.get("n_tau", 16), 5)
    return orc.make_embedding_table(cfg["n_vocab"], dim, cfg["seed"] + off)


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    cfg = json.loads(str(z["cfg"]))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != "cfg"}
    return cfg, arrays


def specs_and_params(cfg):
    n_emb_xyz, n_emb_dir = cfg.get("n_emb_xyz", 10), cfg.get("n_emb_dir", 4)
    spec_c = orc.FieldSpec("coarse", n_emb_xyz=n_emb_xyz, n_emb_dir=n_emb_dir)
    P_c = field_params(spec_c, cfg["seed"], cfg["regime"])
    spec_f = P_f = None
    if cfg["fine"] is not None:
        spec_f = orc.FieldSpec("fine", n_emb_xyz=n_emb_xyz, n_emb_dir=n_emb_dir, n_a=cfg.get("n_a", 48), n_tau=cfg.get("n_tau", 16),
                               encode_appearance=cfg["fine"] in ("a", "at"),
                               encode_transient=cfg["fine"] == "at", beta_min=cfg["beta_min"])
        P_f = field_params(spec_f, cfg["seed"] + 1, cfg["regime"])
    return spec_c, P_c, spec_f, P_f


def latents(cfg, arrays, spec_f):
    a_emb = t_emb = None
    if spec_f is not None and spec_f.encode_appearance:
        a_emb = embedding_table(cfg, "a")[arrays["ts"]]
    if spec_f is not None and spec_f.encode_transient:
        t_emb = embedding_table(cfg, "t")[arrays["ts"]]
    return a_emb, t_emb


def random_inputs(cfg, arrays):
    """Map the captured RNG draws (in the reference's call order, SURVEY.md
    appendix B) onto the injected-randomness arguments."""
    out = dict(perturb_rand=None, noise_coarse=None, u=None, noise_fine=None)
    order = cfg["rng_order"]
    idx = 0
    if cfg["perturb"] > 0:
        assert order[idx] == "rand_like"
        out["perturb_rand"] = arrays[f"rng{idx}_rand_like"]
        idx += 1
    assert order[idx] == "randn_like"
    out["noise_coarse"] = arrays[f"rng{idx}_randn_like"]
    idx += 1
    if cfg["I"] > 0 and cfg["perturb"] > 0:
        assert order[idx] == "rand"
        out["u"] = arrays[f"rng{idx}_rand"]
        idx += 1
    if idx < len(order):
        assert order[idx] == "randn_like"
        out["noise_fine"] = arrays[f"rng{idx}_randn_like"]
        idx += 1
    assert idx == len(order)
    return out


def oracle_kwargs(cfg, arrays):
    spec_c, P_c, spec_f, P_f = specs_and_params(cfg)
    a_emb, t_emb = latents(cfg, arrays, spec_f)
    kw = dict(n_samples=cfg["S"], use_disp=cfg["use_disp"], perturb=cfg["perturb"],
              noise_std=cfg["noise_std"], n_importance=cfg["I"], white_back=cfg["white_back"],
              test_time=cfg["test_time"], a_emb=a_emb, t_emb=t_emb,
              output_transient=True if cfg["output_transient"] is None else cfg["output_transient"])
    kw.update(random_inputs(cfg, arrays))
    if "view_dir" in arrays:
        kw["view_dir"] = arrays["view_dir"]
    if cfg.get("barf_epoch") is not None:
        kw["barf_epoch"] = cfg["barf_epoch"]
        kw["pe_w_xyz"] = orc.barf_weights(spec_c.n_emb_xyz, cfg["barf_epoch"])
        kw["pe_w_dir"] = orc.barf_weights(spec_c.n_emb_dir, cfg["barf_epoch"])
    return (spec_c, P_c, spec_f, P_f), kw


# ----------------------------------------------------------------------------------------------------------------
# Conditioning of the importance sampling of a fixture (which rays an end-to-end comparison can hold to 1e-4)
# ----------------------------------------------------------------------------------------------------------------
SAMPLE_EPS = 1e-5


def sampling_conditioning(z_coarse, w_coarse, u, delta_w=1e-7):
    """Per ray: the largest depth interval within which one of its importance draws may legitimately land when the
    coarse weights are perturbed at the level two correct implementations differ by (returns (R,) tensor, fp64).

    sample_pdf (rendering.py:7-46) is discontinuous in its inputs: `denom[denom < eps] = 1` switches formula for bins
    whose probability is within rounding of eps -- and (w + eps) / sum with eps = 1e-5 puts EVERY empty bin of a nearly
    opaque ray (sum ~ 1) right at that switch, where the float rounding of two cdf entries near 1 (ulp 6e-8..1.2e-7
    against a difference of 1e-5) decides; and searchsorted flips bins at the breakpoints.  Coarse weights that agree
    to 1e-6 (far inside the 1e-4 parity bar) therefore do not imply equal fine depths on such rays: the reference run
    on another device, or with another summation order, differs from itself there.  And the pdf is (w + eps) / sum:
    on a nearly EMPTY ray (sum ~ 1e-3) an absolute weight difference of 6e-8 is a pdf difference of 6e-5.
    This function derives, from the reference's own coarse outputs, how far each ray's draws may move when the
    coarse weights agree to `delta_w` absolute per entry (two fp32-class implementations: ~1e-7):
      delta_cdf = 4 delta_w / sum: uncertainty of a cdf entry relative to u (a few non-cancelling pdf entries),
      delta_den = max(2.5e-7, 2 delta_w / sum): uncertainty of a bin's probability as computed (one pdf entry, and
                  the fp32 rounding of cdf[j+1] - cdf[j] near 1).
    The end-to-end parity test requires every fine depth of the HIP path to lie within this interval of the
    reference's (tests/test_parity_gpu.py); field and compositing are pinned at 1e-4 on ALL rays with the reference's
    depths injected (render_rays(..., z_fine=...)), and the sampler itself bit for bit on identical inputs
    (tests/test_sample_pdf_gpu.py)."""
    zz, ww = z_coarse.double(), (w_coarse[:, 1:-1] + SAMPLE_EPS).double()
    mids = 0.5 * (zz[:, :-1] + zz[:, 1:])
    total = ww.sum(1, keepdim=True)
    pdf = ww / total
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), pdf.cumsum(1)], 1)
    M = pdf.shape[1]
    ud = u.double()
    delta_cdf = 4.0 * delta_w / total                                    # (R, 1)
    delta_den = (2.0 * delta_w / total).clamp(min=2.5e-7)
    j0 = (torch.searchsorted(cdf.contiguous(), ud.contiguous(), right=True) - 1).clamp(min=0)
    lo = torch.full_like(ud, float("inf"))
    hi = torch.full_like(ud, float("-inf"))
    for dj in (-1, 0, 1):
        j = (j0 + dj).clamp(0, M)
        top = j == M
        jn = (j + 1).clamp(max=M)
        c0, c1 = cdf.gather(1, j), cdf.gather(1, jn)
        b0, b1 = mids.gather(1, j), mids.gather(1, jn)
        reach = (ud >= c0 - delta_cdf) & ((ud <= c1 + delta_cdf) | top)
        den = c1 - c0
        width = (b1 - b0).abs()
        frac_num = (ud - c0).clamp(min=0.0)
        for branch in ("keep", "one"):
            if branch == "keep":
                valid = reach & (den >= SAMPLE_EPS - delta_den) & ~top
                d = den.clamp(min=SAMPLE_EPS - delta_den)
            else:
                valid = reach & ((den < SAMPLE_EPS + delta_den) | top)
                d = torch.ones_like(den)
            cand = b0 + (frac_num / d).clamp(max=1.0) * (b1 - b0)
            # cdf[0] is exactly 0 in every implementation: a draw in the first bin has no breakpoint uncertainty
            d_c0 = torch.where(j == 0, torch.zeros_like(den), delta_cdf.expand_as(den))
            spread = (d_c0 + frac_num / d * delta_den).clamp(max=d) / d * width
            lo = torch.where(valid, torch.minimum(lo, cand - spread), lo)
            hi = torch.where(valid, torch.maximum(hi, cand + spread), hi)
    return (hi - lo).clamp(min=0).max(dim=1)[0]


def fixture_conditioning(cfg, arrays, **deltas):
    """sampling_conditioning of a render fixture (None when it has no fine pass), from the reference's own outputs."""
    if cfg["I"] == 0:
        return None
    rays = arrays["rays"]
    rnd = random_inputs(cfg, arrays)
    z = orc.coarse_depths(rays[:, 6:7], rays[:, 7:8], cfg["S"], cfg["use_disp"], cfg["perturb"], rnd["perturb_rand"])
    R, I = rays.shape[0], cfg["I"]
    u = rnd["u"] if cfg["perturb"] > 0 else torch.linspace(0, 1, I).expand(R, I)
    return sampling_conditioning(z, arrays["out.weights_coarse"], u, **deltas)
