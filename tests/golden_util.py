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
    dim, off = (48, 4) if which == "a" else (16, 5)
    return orc.make_embedding_table(cfg["n_vocab"], dim, cfg["seed"] + off)


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    cfg = json.loads(str(z["cfg"]))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != "cfg"}
    return cfg, arrays


def specs_and_params(cfg):
    n_emb_xyz = cfg.get("n_emb_xyz", 10)
    spec_c = orc.FieldSpec("coarse", n_emb_xyz=n_emb_xyz)
    P_c = field_params(spec_c, cfg["seed"], cfg["regime"])
    spec_f = P_f = None
    if cfg["fine"] is not None:
        spec_f = orc.FieldSpec("fine", n_emb_xyz=n_emb_xyz, encode_appearance=cfg["fine"] in ("a", "at"),
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
