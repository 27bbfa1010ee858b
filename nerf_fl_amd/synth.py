"""Seeded synthetic inputs for benchmarks and demos: field parameters and ray batches.

Platform-independent (numpy PCG64), so a run on the GPU box and a run of the CPU checker see the same numbers.
The test oracle keeps its own copies of these generators (it must not depend on the product package);
tests/test_capi_cpu.py asserts that the two stay identical.
"""
import numpy as np
import torch

_W, _H = 256, 128


def field_param_shapes(typ="coarse", n_emb_xyz=10, n_emb_dir=4, encode_appearance=False, n_a=48,
                       encode_transient=False, n_tau=16):
    """Name -> shape of every parameter of a field, in the reference's registration order (models/nerf.py:121-151)."""
    if typ == "coarse":
        encode_appearance = encode_transient = False
    cx, cd = 6 * n_emb_xyz + 3, 6 * n_emb_dir + 3
    shp = {}
    for i in range(8):
        fan_in = cx if i == 0 else (_W + cx if i == 4 else _W)
        shp[f"xyz_encoding_{i + 1}.0.weight"] = (_W, fan_in)
        shp[f"xyz_encoding_{i + 1}.0.bias"] = (_W,)
    shp["xyz_encoding_final.weight"] = (_W, _W)
    shp["xyz_encoding_final.bias"] = (_W,)
    shp["dir_encoding.0.weight"] = (_H, _W + cd + (n_a if encode_appearance else 0))
    shp["dir_encoding.0.bias"] = (_H,)
    shp["static_sigma.0.weight"] = (1, _W)
    shp["static_sigma.0.bias"] = (1,)
    shp["static_rgb.0.weight"] = (3, _H)
    shp["static_rgb.0.bias"] = (3,)
    if encode_transient:
        shp["transient_encoding.0.weight"] = (_H, _W + n_tau)
        shp["transient_encoding.0.bias"] = (_H,)
        for j in (2, 4, 6):
            shp[f"transient_encoding.{j}.weight"] = (_H, _H)
            shp[f"transient_encoding.{j}.bias"] = (_H,)
        for head, n in (("transient_sigma", 1), ("transient_rgb", 3), ("transient_beta", 1)):
            shp[f"{head}.0.weight"] = (n, _H)
            shp[f"{head}.0.bias"] = (n,)
    return shp


def make_field_params(seed, regime="default", **field):
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) weights and biases (nn.Linear's default scale); regime 'sharp' scales the
    density heads (weight x30, bias -3) so that densities are peaky and importance sampling has something to find."""
    shapes = field_param_shapes(**field)
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in shapes.items():
        fan_in = shapes[name.rsplit(".", 1)[0] + ".weight"][1]
        bound = 1.0 / np.sqrt(fan_in)
        out[name] = torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))
    if regime == "sharp":
        for head in ("static_sigma", "transient_sigma"):
            if f"{head}.0.weight" in out:
                out[f"{head}.0.weight"] = out[f"{head}.0.weight"] * 30.0
                out[f"{head}.0.bias"] = out[f"{head}.0.bias"] - 3.0
    elif regime != "default":
        raise ValueError(regime)
    return out


def make_rays(n_rays, seed, near=2.0, far=6.0):
    """Blender-like rays (datasets/blender.py:65-66): origins near (0, 0, 4), unit directions towards the origin."""
    rng = np.random.default_rng(seed)
    o = np.array([0.0, 0.0, 4.0]) + 0.1 * rng.standard_normal((n_rays, 3))
    tgt = 0.8 * rng.uniform(-1, 1, size=(n_rays, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    nf = np.stack([np.full(n_rays, near), np.full(n_rays, far)], axis=1)
    return torch.from_numpy(np.concatenate([o, d, nf], axis=1).astype(np.float32))


def make_rays_photo(n_rays, seed):
    """Phototourism-like rays (datasets/phototourism.py:130-140): scattered cameras, near/far different on every ray,
    largest far bound 5."""
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
