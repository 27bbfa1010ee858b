"""Build nerf_fl_amd modules from the oracle's seeded parameters and call the HIP path."""
import torch

import nerf_fl_amd
from nerf_fl_amd import NeRF, PosEmbedding, render_rays
from oracle import nerfw_oracle as orc

DEV = "cuda:0"


def module_from(spec, P, refine_pose=False):
    m = NeRF(spec.typ, in_channels_xyz=spec.c_xyz, in_channels_dir=spec.c_dir,
             encode_appearance=spec.encode_appearance, in_channels_a=spec.n_a,
             encode_transient=spec.encode_transient, in_channels_t=spec.n_tau, beta_min=spec.beta_min,
             refine_pose=refine_pose)
    m.load_state_dict({k: v.detach().clone() for k, v in P.items()}, strict=True)
    return m.to(DEV)


def make_embeddings(n_emb_xyz, barf, n_emb_dir=4):
    if barf:
        from nerf_fl_amd import BarfPosEmbedding
        return {"xyz": BarfPosEmbedding(n_emb_xyz - 1, n_emb_xyz, 4, 8), "dir": BarfPosEmbedding(n_emb_dir - 1, n_emb_dir, 4, 8)}
    return {"xyz": PosEmbedding(n_emb_xyz - 1, n_emb_xyz), "dir": PosEmbedding(n_emb_dir - 1, n_emb_dir)}


def hip_render(specs, rays, kw, precision="f16x3", field_raw=False):
    """kw: the oracle_kwargs dict of golden_util (same semantics as oracle.render_rays)."""
    spec_c, P_c, spec_f, P_f = specs
    nerf_fl_amd.set_precision(precision)
    barf = kw.get("pe_w_xyz") is not None
    models = {"coarse": module_from(spec_c, P_c, barf)}
    if spec_f is not None:
        models["fine"] = module_from(spec_f, P_f, barf)
    emb = make_embeddings(spec_c.n_emb_xyz, barf, spec_c.n_emb_dir)
    extra = {}
    if barf:
        extra["current_epoch"] = kw["barf_epoch"]
    for k in ("perturb_rand", "noise_coarse", "u", "noise_fine"):
        if kw.get(k) is not None:
            extra[k] = kw[k].to(DEV)
    if kw.get("a_emb") is not None:
        extra["a_embedded"] = kw["a_emb"].to(DEV)
    if kw.get("t_emb") is not None:
        extra["t_embedded"] = kw["t_emb"].to(DEV)
    if kw.get("view_dir") is not None:
        extra["view_dir"] = kw["view_dir"].to(DEV)
    if kw.get("z_fine") is not None:       # fine depths of the reference run (fixture array `z_fine`)
        extra["z_fine"] = kw["z_fine"].to(DEV)
    if not kw.get("output_transient", True):
        extra["output_transient"] = False
    if field_raw:
        extra["_field_raw"] = True
    ts = torch.zeros(rays.shape[0], dtype=torch.long, device=DEV)
    with torch.no_grad():
        res = render_rays(models, emb, rays.to(DEV), ts, kw["n_samples"], kw["use_disp"], kw["perturb"],
                          kw["noise_std"], kw["n_importance"], 32768, kw["white_back"], kw["test_time"], **extra)
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in res.items()}
