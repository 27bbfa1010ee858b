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
    """One fixed-shape render_rays(test_time=True) call captured in a HIP graph.

    Everything that varies between replays lives in static device buffers the captured kernels read: the ray matrix (or,
    for `CameraRays` input, the camera struct: the render kernel's prologue generates the rays from it -- C ABI
    nfl_pass_args::d_cam), `ts`, and the per-ray kwargs `a_embedded` / `t_embedded` / `view_dir`.  The packed weight
    streams are NOT part of the graph: `__call__` compares the fields' parameter keys (data pointers + version counters)
    with the ones the streams were packed from and re-packs eagerly before the replay when they moved (an optimizer
    step, load_state_dict), so a cached graph never renders stale weights."""

    PER_RAY = ("a_embedded", "t_embedded", "view_dir")

    def __init__(self, models, embeddings, chunk, device, N_samples, use_disp, N_importance, white_back, camera=False,
                 **kwargs):
        import weakref

        from . import rendering as rnd
        self.chunk, self.device = chunk, torch.device(device)
        self.camera = bool(camera)
        self.model_refs = {k: weakref.ref(m) for k, m in models.items()}
        self.rays = torch.zeros(chunk, 8, device=device)
        self.rays[:, 3:6] = torch.tensor([0.0, 0.0, -1.0], device=device)
        self.rays[:, 6], self.rays[:, 7] = 2.0, 6.0
        self.ts = torch.zeros(chunk, dtype=torch.long, device=device)
        self.kw_static = {}
        for k in self.PER_RAY:       # per-frame latent overrides (test_phototourism.ipynb cell 11), per-ray view directions
            if kwargs.get(k) is not None:
                v = kwargs[k].to(device=device, dtype=torch.float32)
                self.kw_static[k] = v[:1].expand(chunk, -1).contiguous().clone()
        other = {k: v for k, v in kwargs.items() if k not in self.PER_RAY}
        self.cam_rays = None
        if self.camera:             # a 1 x 1 frame as a placeholder; __call__ overwrites the device-side struct
            self.cam_rays = CameraRays(torch.eye(4)[:3], torch.eye(3), 1, 1, 2.0, 6.0, device, count=0)
            self.cam_rays.count, self.cam_rays.shape = chunk, (chunk, 8)
            self.cam_rays.to_device()

        def run():
            return render_rays(models, embeddings, self.cam_rays if self.camera else self.rays, self.ts, N_samples, use_disp,
                               0, 0, N_importance, chunk, white_back, True, **self.kw_static, **other)

        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():      # warm-up: packs weights, sets kernel attributes
            run()
        torch.cuda.current_stream(device).wait_stream(side)
        n_xyz, n_dir = rnd._n_freqs(embeddings["xyz"]), rnd._n_freqs(embeddings["dir"])
        self.fields = [rnd._field(m, n_xyz, n_dir, self.device, pack=False) for m in models.values()]
        self.graph = torch.cuda.CUDAGraph()
        import torch.distributed as dist
        # a process group's watchdog thread may query events while we capture: confine the capture checks to this thread
        mode = dict(capture_error_mode="thread_local") if dist.is_available() and dist.is_initialized() else {}
        with torch.no_grad(), torch.cuda.graph(self.graph, **mode):
            self.out = run()

    def alive_for(self, models):
        """The captured launches read THESE models' packed streams: a cache hit must be for the same module objects."""
        return set(models) == set(self.model_refs) and all(self.model_refs[k]() is m for k, m in models.items())

    def __call__(self, rays, ts, **per_ray):
        from . import rendering as rnd
        n = rays.shape[0]
        if n > self.chunk:
            raise ValueError(f"GraphedChunk captured for {self.chunk} rays, got {n}")
        if isinstance(rays, CameraRays):
            if self.camera:          # 88 bytes to the device-side camera struct the captured prologue reads
                self.cam_rays.load(rays)
            else:                    # captured on a ray matrix: materialise the rows with nfl_gen_rays
                rays = frame_rays_of(rays)
        elif self.camera:
            raise ValueError("this GraphedChunk was captured for CameraRays input")
        if not isinstance(rays, CameraRays):
            self.rays[:n].copy_(rays)
            if n < self.chunk:                                   # ragged tail: repeat the last ray
                self.rays[n:].copy_(rays[-1:].expand(self.chunk - n, -1))
        if ts is not None:
            self.ts[:n].copy_(ts)
            if n < self.chunk:
                self.ts[n:].copy_(ts[-1:].expand(self.chunk - n))
        if set(per_ray) - set(self.kw_static):
            raise ValueError(f"per-ray kwargs {sorted(set(per_ray) - set(self.kw_static))} were not part of the capture")
        for k, buf in self.kw_static.items():
            if k not in per_ray:
                raise ValueError(f"this GraphedChunk was captured with `{k}`: pass it on every call")
            v = per_ray[k]
            if v.shape[0] == 1:
                buf.copy_(v.expand(self.chunk, -1))
            else:                                            # per-ray values of a (possibly ragged) chunk
                buf[:n].copy_(v)
                if n < self.chunk:
                    buf[n:].copy_(v[-1:].expand(self.chunk - n, -1))
        # weights moved since the streams were packed (optimizer step, load_state_dict)?  re-pack eagerly: the captured
        # kernels read the same buffers
        rnd._pack_streams(self.fields, bwd=False, rays_grad=False)
        self.graph.replay()
        return {k: v[:n] for k, v in self.out.items()}


_GRAPH_CACHE = {}      # default cache of batched_inference(use_graph=True): one capture per distinct launch sequence


def _graph_key(models, embeddings, rays, chunk, N_samples, N_importance, use_disp, white_back, kwargs):
    """Everything that changes the captured launch sequence or the buffers it reads."""
    from . import rendering as rnd
    return (chunk, N_samples, N_importance, bool(use_disp), bool(white_back), str(rays.device), isinstance(rays, CameraRays),
            tuple(sorted((k, id(m)) for k, m in models.items())), tuple(sorted((k, id(e)) for k, e in embeddings.items())),
            tuple(sorted(k for k in kwargs if kwargs[k] is not None and k in GraphedChunk.PER_RAY)),
            tuple(sorted((k, repr(v)) for k, v in kwargs.items() if k not in GraphedChunk.PER_RAY)),
            rnd.get_precision())


@torch.no_grad()
def batched_inference(models, embeddings, rays, ts, N_samples, N_importance, use_disp=False, chunk=1024 * 128,
                      white_back=False, use_graph=False, _graph_cache=None, **kwargs):
    """Same arguments as the reference's eval.batched_inference (eval.py:80-88) plus `use_graph`.
    Returns a dict of GPU tensors covering all `rays`.  With `use_graph` the chunk is captured once per distinct call
    signature (models, kwargs present, `output_transient`, shapes, precision) and kept in `_graph_cache` (default: a
    module-level cache), so repeated calls replay instead of re-capturing."""
    B = rays.shape[0]
    results = {}
    runner = None
    if use_graph:
        key = _graph_key(models, embeddings, rays, chunk, N_samples, N_importance, use_disp, white_back, kwargs)
        cache = _graph_cache if _graph_cache is not None else _GRAPH_CACHE
        if key not in cache or not cache[key].alive_for(models):
            cache[key] = GraphedChunk(models, embeddings, chunk, rays.device, N_samples, use_disp, N_importance,
                                      white_back, camera=isinstance(rays, CameraRays), **kwargs)
        runner = cache[key]
    for i in range(0, B, chunk):
        r = rays.slice(i, i + chunk) if isinstance(rays, CameraRays) else rays[i:i + chunk]
        t = ts[i:i + chunk] if ts is not None else None         # reference eval.py:94
        n = r.shape[0]
        # per-ray kwargs follow their chunk: a (B, C) tensor is sliced, a (1, C) one (one latent code for the whole
        # frame, test_phototourism.ipynb cell 11) is broadcast
        per_ray = {}
        for k in GraphedChunk.PER_RAY:
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
    cache = kwargs.pop("_graph_cache", None)        # None: the module-level cache of batched_inference
    frames = [render_frame(models, embeddings, poses[i], K, H, W, near, far, N_samples, N_importance, _graph_cache=cache,
                           **kwargs)[0] for i in range(lo, hi)]
    dev = kwargs.get("device", "cuda:0")
    return lo, (torch.stack(frames) if frames else torch.empty(0, H, W, 3, dtype=torch.uint8, device=dev))
