"""Throughput of the other BASELINE.json configs (not the bench.py headline): configs[2] NeRF-W train
step, configs[4]-like eval chunk (128+128, test_time) direct and HIP-graph replayed.  GPU box only."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_fl_amd
from nerf_fl_amd import NeRF, PosEmbedding, render_rays
from nerf_fl_amd.eval import batched_inference
from nerf_fl_amd.train import NerfWLoss
from oracle import nerfw_oracle as orc

dev = torch.device("cuda", 0)
nerf_fl_amd.set_precision("f16x3")


def models_for(a, t, n_vocab):
    sc = orc.FieldSpec("coarse")
    sf = orc.FieldSpec("fine", encode_appearance=a, encode_transient=t, beta_min=0.1)
    m = {"coarse": NeRF("coarse").to(dev), "fine": NeRF("fine", encode_appearance=a, encode_transient=t, beta_min=0.1).to(dev)}
    m["coarse"].load_state_dict(orc.make_field_params(sc, 11, "sharp"))
    m["fine"].load_state_dict(orc.make_field_params(sf, 12, "sharp"))
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    if a:
        emb["a"] = torch.nn.Embedding(n_vocab, 48).to(dev)
    if t:
        emb["t"] = torch.nn.Embedding(n_vocab, 16).to(dev)
    return m, emb


def timed(fn, steps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


out = {}
# ---- configs[2]: full NeRF-W (a + t), 4096 rays, 64+64, train step
R = 4096
models, emb = models_for(True, True, 100)
params = [p for m in list(models.values()) + [emb["a"], emb["t"]] for p in m.parameters()]
from nerf_fl_amd.train import Adam
opt = Adam(params, lr=5e-4, eps=1e-8)
rays, ts = orc.make_rays(R, 100).to(dev), torch.randint(0, 100, (R,), device=dev)
target, loss_fn = torch.rand(R, 3, device=dev), NerfWLoss()

def train_step():
    opt.zero_grad(set_to_none=True)
    render_rays(models, emb, rays, ts, 64, False, 1.0, 1.0, 64, 32768, True, False, loss_target=target)["_nerfw_loss"].backward()
    opt.step()

dt = timed(train_step, 30, 8)
out["cfg3_nerfw_train"] = {"ms_per_step": dt * 1e3, "ray_samples_per_s": R * 128 / dt}

# ---- configs[4]-like: eval chunk, 128+128, test_time, NeRF-W; chunk 32768 rays x 4
R, chunk = 131072, 32768
rays, ts = orc.make_rays(R, 101, near=0.3, far=5.0).to(dev), torch.randint(0, 100, (R,), device=dev)
cache = {}
for name, graph in (("direct", False), ("hip_graph", True)):
    dt = timed(lambda: batched_inference(models, emb, rays, ts, 128, 128, chunk=chunk, white_back=False,
                                         use_graph=graph, _graph_cache=cache), 5, 2)
    out[f"cfg5_eval_{name}"] = {"ms_per_131072_rays": dt * 1e3, "ray_samples_per_s": R * 256 / dt,
                                "rays_per_s": R / dt}
# ---- one 800 x 800 frame from (pose, intrinsics): rays generated in the kernel prologue vs a materialised ray matrix
from nerf_fl_amd.eval import fov60_intrinsics, frame_rays, render_frame
from nerf_fl_amd.poses import make_c2w
H = W = 800
K = fov60_intrinsics(W, H)
c2w = make_c2w(torch.tensor([0.0, 0.05, 0.0]), torch.tensor([0.0, 0.0, 4.0]))[:3]
kw = dict(chunk=chunk, white_back=False, device=dev, output_transient=False)
dt = timed(lambda: render_frame(models, emb, c2w, K, H, W, 0.3, 5.0, 128, 128, ts=7, **kw), 3, 1)
out["cfg5_frame_800x800_camera_prologue"] = {"ms_per_frame": dt * 1e3, "rays_per_s": H * W / dt}
# the same with the chunk replayed from ONE HIP graph whose prologue reads the camera from device memory (configs[4]:
# "hipGraph-captured eval.py inference" + the camera prologue, combined since round 3: nfl_pass_args::d_cam)
gcache = {}
dt = timed(lambda: render_frame(models, emb, c2w, K, H, W, 0.3, 5.0, 128, 128, ts=7, use_graph=True, _graph_cache=gcache, **kw), 3, 1)
out["cfg5_frame_800x800_camera_prologue_hip_graph"] = {"ms_per_frame": dt * 1e3, "rays_per_s": H * W / dt, "captures": len(gcache)}
ts_f = torch.full((H * W,), 7, dtype=torch.long, device=dev)
dt = timed(lambda: batched_inference(models, emb, frame_rays(c2w, K, H, W, 0.3, 5.0, dev), ts_f, 128, 128, chunk=chunk,
                                     white_back=False, output_transient=False), 3, 1)
out["cfg5_frame_800x800_ray_matrix"] = {"ms_per_frame": dt * 1e3, "rays_per_s": H * W / dt}
print(json.dumps(out, indent=1))
