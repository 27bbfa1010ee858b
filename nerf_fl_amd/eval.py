"""Chunked no-grad inference for whole images / video frames (reference eval.py:80-110).

`batched_inference` mirrors the reference helper of the same name: it walks the rays in
chunks of `chunk` rays, calls render_rays(..., test_time=True) on each and concatenates the
per-key results.  Differences that matter on MI355X:

  * results stay on the GPU (the reference moves every chunk to the host, eval.py:106);
  * every chunk has the same shape -- the last, ragged one is padded with copies of its final
    ray and the padding is dropped on write-back -- so one chunk can be captured ONCE into a
    HIP graph (`use_graph=True`) and replayed: a frame becomes a handful of graph launches
    with no per-kernel host work (config 5: 128+128 samples, chunk 131072).
"""
import torch

from . import parallel
from .rendering import CameraRays, render_rays

__all__ = ["batched_inference", "GraphedChunk", "frame_rays", "to_uint8", "dolly_path", "render_frame", "render_video"]


class GraphedChunk:
    """One fixed-shape render_rays(test_time=True) call captured in a HIP graph."""

    def __init__(self, models, embeddings, chunk, device, N_samples, use_disp, N_importance, white_back, **kwargs):
        self.chunk = chunk
        self.rays = torch.zeros(chunk, 8, device=device)
        self.rays[:, 3:6] = torch.tensor([0.0, 0.0, -1.0], device=device)
        self.rays[:, 6], self.rays[:, 7] = 2.0, 6.0
        self.ts = torch.zeros(chunk, dtype=torch.long, device=device)
        self.kw_static = {}
        for k in ("a_embedded", "t_embedded"):       # per-frame latent overrides (test_phototourism.ipynb cell 11)
            if kwargs.get(k) is not None:
                self.kw_static[k] = kwargs[k].expand(chunk, -1).contiguous().clone()
        other = {k: v for k, v in kwargs.items() if k not in self.kw_static}

        def run():
            return render_rays(models, embeddings, self.rays, self.ts, N_samples, use_disp, 0, 0, N_importance,
                               chunk, white_back, True, **self.kw_static, **other)

        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():      # warm-up: packs weights, sets kernel attributes
            run()
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = run()

    def __call__(self, rays, ts, **latents):
        n = rays.shape[0]
        if isinstance(rays, CameraRays):      # a captured launch freezes by-value arguments (the camera): materialise
            rays = frame_rays_of(rays)        # the rows into the static buffer with nfl_gen_rays instead
        self.rays[:n].copy_(rays)
        if ts is not None:
            self.ts[:n].copy_(ts)
        if n < self.chunk:                                   # ragged tail: repeat the last ray
            self.rays[n:].copy_(rays[-1:].expand(self.chunk - n, -1))
            if ts is not None:
                self.ts[n:].copy_(ts[-1:].expand(self.chunk - n))
        for k, buf in self.kw_static.items():
            if k in latents:
                v = latents[k]
                if v.shape[0] == 1:
                    buf.copy_(v.expand(self.chunk, -1))
                else:                                        # per-ray values of a (possibly ragged) chunk
                    buf[:n].copy_(v)
                    if n < self.chunk:
                        buf[n:].copy_(v[-1:].expand(self.chunk - n, -1))
        self.graph.replay()
        return {k: v[:n] for k, v in self.out.items()}


@torch.no_grad()
def batched_inference(models, embeddings, rays, ts, N_samples, N_importance, use_disp=False, chunk=1024 * 128,
                      white_back=False, use_graph=False, _graph_cache=None, **kwargs):
    """Same arguments as the reference's eval.batched_inference (eval.py:80-88) plus `use_graph`.
    Returns a dict of GPU tensors covering all `rays`."""
    B = rays.shape[0]
    results = {}
    runner = None
    if use_graph:
        key = (chunk, N_samples, N_importance, bool(use_disp), bool(white_back), str(rays.device))
        cache = _graph_cache if _graph_cache is not None else {}
        if key not in cache:
            cache[key] = GraphedChunk(models, embeddings, chunk, rays.device, N_samples, use_disp, N_importance,
                                      white_back, **kwargs)
        runner = cache[key]
    for i in range(0, B, chunk):
        r = rays.slice(i, i + chunk) if isinstance(rays, CameraRays) else rays[i:i + chunk]
        t = ts[i:i + chunk] if ts is not None else None         # reference eval.py:94
        n = r.shape[0]
        # per-ray kwargs follow their chunk: a (B, C) tensor is sliced, a (1, C) one (one latent code for the whole
        # frame, test_phototourism.ipynb cell 11) is broadcast
        per_ray = {}
        for k in ("a_embedded", "t_embedded", "view_dir"):
            v = kwargs.get(k)
            if v is not None:
                per_ray[k] = v[i:i + chunk] if v.shape[0] == B and B != 1 else v.expand(n, -1)
        if runner is not None:
            out = {k: v.clone() for k, v in runner(r, t, **per_ray).items()}
        else:
            out = render_rays(models, embeddings, r, t, N_samples, use_disp, 0, 0, N_importance, chunk, white_back,
                              True, **{**kwargs, **per_ray})
        for k, v in out.items():
            results.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0) for k, v in results.items()}


def frame_rays(c2w, K, H, W, near, far, device, start=0, count=None):
    """(count, 8) ray matrix [o, d, near, far] of pixels [start, start + count) of an H x W frame seen from pose `c2w`
    (3|4, 4) with intrinsics `K` (3, 3), generated on the device by nfl_gen_rays (reference datasets/ray_utils.py:5-55
    + the near/far columns the datasets append, e.g. blender.py:65-69).  Only the 12 pose floats cross the bus."""
    import ctypes as C

    from . import _lib
    count = H * W - start if count is None else int(count)
    if start < 0 or count < 0 or start + count > H * W:
        raise ValueError("pixel range outside the frame")
    c2w = torch.as_tensor(c2w, dtype=torch.float32).cpu()[:3, :4].contiguous()
    K = torch.as_tensor(K, dtype=torch.float32).cpu()
    pose = (C.c_float * 12)(*c2w.reshape(-1).tolist())
    dev = torch.device(device)
    rays = torch.empty(count, 8, dtype=torch.float32, device=dev)
    if count == 0:
        return rays
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().nfl_gen_rays(pose, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), int(W),
                                           int(start), count, float(near), float(far), C.c_void_p(rays.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nfl_gen_rays")
    return rays


def frame_rays_of(cam_rays):
    """The (count, 8) matrix a CameraRays stands for."""
    c = cam_rays.cam
    c2w = torch.tensor(list(c.c2w), dtype=torch.float32).reshape(3, 4)
    K = torch.tensor([[c.fx, 0.0, c.cx], [0.0, c.fy, c.cy], [0.0, 0.0, 1.0]])
    return frame_rays(c2w, K, cam_rays.H, cam_rays.W, c.near, c.far, cam_rays.device, cam_rays.start, cam_rays.count)


def to_uint8(rgb):
    """clip to [0, 1], x 255, truncate -- the reference's image conversion (eval.py:201-203: np.clip(...), (img * 255)
    .astype(np.uint8)) -- on the device, so a frame leaves the GPU as 3 bytes per pixel instead of 12."""
    return (rgb.clamp(0.0, 1.0) * 255.0).to(torch.uint8)


def dolly_path(c2w0, n_frames=120, dx=0.03, dy=-0.1, dz=0.5):
    """Camera path of the reference's novel-view video (eval.py:169-183): `n_frames` copies of a training pose whose
    translation moves linearly by (dx, dy, dz) -- the reference hard-codes exactly this for brandenburg_gate (pose of
    image 1123, 30 * 4 frames, 0.03 / -0.1 / 0.5) and raises NotImplementedError for every other scene; here it is the
    path of any scene (configs[4], trevi_fountain: choose the start pose and the deltas).  Returns (n_frames, 3, 4)."""
    c2w0 = torch.as_tensor(c2w0, dtype=torch.float32)[:3, :4]
    poses = c2w0[None].repeat(n_frames, 1, 1)
    for axis, d in enumerate((dx, dy, dz)):
        poses[:, axis, 3] += torch.linspace(0, d, n_frames)
    return poses


def fov60_intrinsics(W, H):
    """The reference's test camera (eval.py:164-168): fov 60 degrees, principal point at the image centre."""
    import math
    f = W / 2 / math.tan(math.pi / 6)
    return torch.tensor([[f, 0.0, W / 2], [0.0, f, H / 2], [0.0, 0.0, 1.0]])


@torch.no_grad()
def render_frame(models, embeddings, c2w, K, H, W, near, far, N_samples, N_importance, ts=None, use_disp=False,
                 chunk=1024 * 128, white_back=False, device="cuda:0", use_graph=False, _graph_cache=None, **kwargs):
    """One H x W frame from (pose, intrinsics): the rays are generated in the render kernel's prologue (CameraRays), the
    image is converted on the device.  Returns (uint8 (H, W, 3), dict of the float outputs).  `ts`: image id for the
    latent tables -- an int, a (H*W,) tensor, or None when `a_embedded` is given / the model has no latent inputs."""
    cam = CameraRays(c2w, K, H, W, near, far, device)
    if isinstance(ts, int):
        ts = torch.full((H * W,), ts, dtype=torch.long, device=cam.device)
    res = batched_inference(models, embeddings, cam, ts, N_samples, N_importance, use_disp, chunk, white_back,
                            use_graph=use_graph, _graph_cache=_graph_cache, **kwargs)
    return to_uint8(res["rgb_fine" if "rgb_fine" in res else "rgb_coarse"]).view(H, W, 3), res


@torch.no_grad()
def render_video(models, embeddings, poses, K, H, W, near, far, N_samples, N_importance, rank=0, world=1, **kwargs):
    """Frames of a camera path, sharded over ranks with no collective: rank r renders the contiguous block
    parallel.shard_bounds(n_frames, r, world) and returns (first frame index, uint8 (n_local, H, W, 3)).  (The
    reference renders every frame on one GPU, eval.py:186-194.)"""
    lo, hi = parallel.shard_bounds(len(poses), rank, world)
    cache = kwargs.pop("_graph_cache", {})
    frames = [render_frame(models, embeddings, poses[i], K, H, W, near, far, N_samples, N_importance, _graph_cache=cache,
                           **kwargs)[0] for i in range(lo, hi)]
    dev = kwargs.get("device", "cuda:0")
    return lo, (torch.stack(frames) if frames else torch.empty(0, H, W, 3, dtype=torch.uint8, device=dev))
