"""A thin training harness for the HIP renderer (the role of the reference's NeRFSystem +
Lightning Trainer, train.py:33-241, without Lightning).

What it keeps from the reference:
  * modules and their checkpoint key prefixes -- `nerf_coarse.*`, `nerf_fine.*`, `embedding_a.*`,
    `embedding_t.*` (train.py:51-76) -- so `utils.load_ckpt` (utils/__init__.py:67-88) and this
    harness can exchange weights;
  * Adam(lr, eps=1e-8) (utils/__init__.py:30-32), optional cosine / step decay per epoch
    (utils/__init__.py:44-61), NerfWLoss (losses.py:18-50), PSNR = -10 log10(mse) (metrics.py:12-13).
What it does differently, because the renderer is ~10^3 x faster than the data path around it:
  * rays / colours / image ids live on the GPU as flat tensors; a batch is a slice of a device-side
    permutation (no DataLoader workers, no per-item collation: train.py:144-149);
  * rays are the upstream 8-column Blender contract (o, d, near, far) handed straight to render_rays;
  * multi-GPU: every rank shuffles its own shard; gradients are averaged with one flat all-reduce.
"""
import math
import os

import torch
from torch import nn

from . import parallel
from .nerf import NeRF, PosEmbedding
from .rendering import render_rays

__all__ = ["NerfWLoss", "psnr", "Adam", "RayTrainer", "GraphedTrainStep"]


class NerfWLoss(nn.Module):
    """Equation 13 of NeRF-W as the reference implements it (losses.py:18-50): c_l, f_l, b_l (+3), s_l."""

    def __init__(self, coef=1.0, lambda_u=0.01):
        super().__init__()
        self.coef, self.lambda_u = coef, lambda_u

    def forward(self, inputs, targets):
        """The terms through the C ABI (nfl_loss_forward / nfl_loss_backward): two launches instead of ~16.  Device tensors
        only, like everything in this package (the CPU restatement of the loss is oracle/nerfw_oracle.py: nerfw_loss)."""
        if not targets.is_cuda:
            raise RuntimeError("nerf_fl_amd.train.NerfWLoss: device tensors only (this build has no CPU path)")
        rgb_f, beta = inputs.get("rgb_fine"), inputs.get("beta")
        tsig = inputs.get("transient_sigmas") if beta is not None else None
        c_l, f_l, b_l, s_l = _FusedNerfWLoss.apply(inputs["rgb_coarse"], rgb_f, beta, tsig, targets, float(self.coef),
                                                   float(self.lambda_u))
        ret = {"c_l": c_l}
        if rgb_f is not None:
            ret["f_l"] = f_l
            if beta is not None:
                ret["b_l"], ret["s_l"] = b_l, s_l
        return ret


class _FusedNerfWLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_c, rgb_f, beta, tsig, targets, coef, lambda_u):
        import ctypes as C

        from . import _lib
        ctx.set_materialize_grads(False)
        f32c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        rgb_c, rgb_f, beta, tsig, targets = f32c(rgb_c), f32c(rgb_f), f32c(beta), f32c(tsig), f32c(targets)
        a = _lib.LossArgs()
        ptr = lambda t: None if t is None else t.data_ptr()
        a.d_rgb_coarse, a.d_rgb_fine, a.d_beta, a.d_transient_sigmas, a.d_target = ptr(rgb_c), ptr(rgb_f), ptr(beta), ptr(tsig), ptr(targets)
        a.n_rays, a.n_samples = rgb_c.shape[0], (tsig.shape[1] if tsig is not None else 0)
        a.coef, a.lambda_u = coef, lambda_u
        losses = torch.empty(4, dtype=torch.float32, device=targets.device)
        a.d_losses = losses.data_ptr()
        with torch.cuda.device(targets.device):
            _lib.check(_lib.lib().nfl_loss_forward(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nfl_loss_forward")
        ctx.saved = (rgb_c, rgb_f, beta, tsig, targets, coef, lambda_u)
        return losses[0], losses[1], losses[2], losses[3]

    @staticmethod
    def backward(ctx, *go):
        import ctypes as C

        from . import _lib
        rgb_c, rgb_f, beta, tsig, targets, coef, lambda_u = ctx.saved
        a = _lib.LossArgs()
        ptr = lambda t: None if t is None else t.data_ptr()
        a.d_rgb_coarse, a.d_rgb_fine, a.d_beta, a.d_transient_sigmas, a.d_target = ptr(rgb_c), ptr(rgb_f), ptr(beta), ptr(tsig), ptr(targets)
        a.n_rays, a.n_samples = rgb_c.shape[0], (tsig.shape[1] if tsig is not None else 0)
        a.coef, a.lambda_u = coef, lambda_u
        keep = [None if g is None else g.to(torch.float32).contiguous() for g in go]
        for k in range(4):
            a.d_grad_loss[k] = ptr(keep[k])
        g_c = torch.empty_like(rgb_c)
        g_f = torch.empty_like(rgb_f) if rgb_f is not None else None
        g_b = torch.empty_like(beta) if beta is not None else None
        g_s = torch.empty_like(tsig) if tsig is not None else None
        a.d_g_rgb_coarse, a.d_g_rgb_fine, a.d_g_beta, a.d_g_transient_sigmas = ptr(g_c), ptr(g_f), ptr(g_b), ptr(g_s)
        with torch.cuda.device(targets.device):
            _lib.check(_lib.lib().nfl_loss_backward(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nfl_loss_backward")
        return g_c, g_f, g_b, g_s, None, None, None


def psnr(pred, gt):
    return -10.0 * torch.log10(((pred - gt) ** 2).mean())


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas, eps) -- no weight decay, no amsgrad, the reference's settings
    (utils/__init__.py:30-32) -- with the whole step in ONE kernel launch (C ABI `nfl_adam_step`).  A regular
    torch Optimizer otherwise: param_groups (LR schedulers work), state[p] = {step, exp_avg, exp_avg_sq} with
    torch's names, so state_dict()s are interchangeable with torch.optim.Adam's."""

    def __init__(self, params, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        """capturable=True: learning rate, betas, eps and the step count are kept in device memory and read by the
        kernel (C ABI `nfl_adam_step_dev`), so `step()` can be captured in a HIP graph and replayed while a scheduler
        changes the rate (GraphedTrainStep); `sync_hyper()` uploads the current param_groups' values.
        The device-side step count is ONE counter per (param group, device), seeded from the largest host-side step of
        the group: all tensors of a group share their bias corrections (torch.optim.Adam counts per parameter; the two
        agree whenever every parameter of a group receives a gradient at every step, which is how this package trains)."""
        self._dev = {}            # (group index, device) -> dict(hyper=float[4] tensor, step=int32 tensor, host=tuple)
        self._captured = set()    # id(p) of the parameters a captured step() updates
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = bool(capturable)

    def _dev_state(self, gi, group, dev):
        k = (gi, str(dev))
        if k not in self._dev:
            steps = [int(self.state[p]["step"]) for p in group["params"] if self.state.get(p)]
            self._dev[k] = dict(hyper=torch.zeros(4, dtype=torch.float32, device=dev),
                                step=torch.full((1,), max(steps) if steps else 0, dtype=torch.int32, device=dev), host=None)
        return self._dev[k]

    def load_state_dict(self, state_dict):
        """torch's loader, then the device-side step counters of a capturable optimizer are dropped so that the next
        step() re-seeds them from the loaded state (bias corrections follow the checkpoint, not the steps this object took
        before)."""
        super().load_state_dict(state_dict)
        self._dev = {}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if getattr(self, "_dev", None):
            self._dev = {}

    def sync_hyper(self):
        """Upload lr / betas / eps of every param group to the device copies the captured launches read (host -> device
        copies: call it outside graph capture; GraphedTrainStep.replay() does)."""
        for (gi, _dev), st in self._dev.items():
            g = self.param_groups[gi]
            host = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
            if st["host"] != host:
                st["hyper"].copy_(torch.tensor(host, dtype=torch.float32))
                st["host"] = host

    def note_replay(self):
        """A captured step() was replayed: advance the host-side step counts (state_dict compatibility) of the parameters
        the captured launch updates (those that had a gradient when it was captured)."""
        for group in self.param_groups:
            for p in group["params"]:
                if self.state.get(p) and (not self._captured or id(p) in self._captured):
                    self.state[p]["step"] = int(self.state[p]["step"]) + 1

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes as C

        from . import _lib
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        capturing = self.capturable and torch.cuda.is_current_stream_capturing()
        for gi, group in enumerate(self.param_groups):
            if self.capturable:                  # device-side counters start from the steps ALREADY taken
                for dev in {p.device for p in group["params"] if p.grad is not None}:
                    self._dev_state(gi, group, dev)
            todo = {}                            # (device, step) -> list of (p, grad, m, v)
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse:
                    raise RuntimeError("nerf_fl_amd.train.Adam: dense fp32 parameters on a ROCm device only")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if not capturing:                # a captured launch runs at replay time: note_replay() counts it
                    st["step"] = int(st["step"]) + 1
                else:
                    self._captured.add(id(p))
                if not p.is_contiguous():
                    raise RuntimeError("nerf_fl_amd.train.Adam: parameters must be contiguous")
                # capturable: one device-side counter per (group, device), so all of a group's tensors step together
                todo.setdefault((p.device, 0 if self.capturable else st["step"]), []).append(
                    (p, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"]))
            b1, b2 = group["betas"]
            for (dev, step), items in todo.items():
                with torch.cuda.device(dev):
                    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                    if self.capturable:
                        ds = self._dev_state(gi, group, dev)
                        if not capturing:
                            self.sync_hyper()
                        elif ds["host"] is None:
                            raise RuntimeError("nerf_fl_amd.train.Adam: run one eager step() (or sync_hyper()) before capture")
                    for i0 in range(0, len(items), _lib.NFL_ADAM_MAX_TENSORS):
                        chunk = items[i0:i0 + _lib.NFL_ADAM_MAX_TENSORS]
                        t = _lib.AdamTensors()
                        for k, (p, g, m, v) in enumerate(chunk):
                            t.param[k], t.grad[k], t.exp_avg[k], t.exp_avg_sq[k] = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                            t.numel[k] = p.numel()
                        if self.capturable:
                            last = i0 + _lib.NFL_ADAM_MAX_TENSORS >= len(items)
                            _lib.check(L.nfl_adam_step_dev(C.byref(t), len(chunk), C.c_void_p(ds["hyper"].data_ptr()),
                                                           C.c_void_p(ds["step"].data_ptr()), int(last), stream),
                                       "nfl_adam_step_dev")
                        else:
                            _lib.check(L.nfl_adam_step(C.byref(t), len(chunk), float(group["lr"]), float(b1), float(b2),
                                                       float(group["eps"]), step, stream), "nfl_adam_step")
                        # the kernel wrote the parameters behind autograd's back: bump their version counters, which
                        # is what tells render_rays to re-pack the weight streams (and autograd to refuse stale graphs)
                        torch.autograd.graph.increment_version([p for p, _, _, _ in chunk])
        return loss


class GraphedTrainStep:
    """One fixed-shape optimisation step -- weight re-pack, render_rays forward, NerfWLoss, the HIP backward, Adam --
    captured ONCE into a HIP graph and replayed: a step becomes one graph launch (~35 kernel launches, ~25 allocations
    and their Python disappear from the host's critical path; what matters at the README batch of 1024 rays, where the
    kernels take ~1.3 ms).  Random draws come from torch's graph-safe Philox generator, so every replay draws afresh.

    `opt` must be `Adam(..., capturable=True)`; `loss_fn=None` uses the loss fused into the render kernels.  Batches are
    loaded into the static buffers with `load()`.
    Construction runs `warmup` REAL steps eagerly (kernel attributes, optimizer state) before capturing.
    With `all_reduce=True` (ranks > 1) the step is two graphs with the flat gradient all-reduce between them."""

    def __init__(self, models, embeddings, params, opt, loss_fn, rays, ts, target, N_samples, N_importance,
                 use_disp=False, perturb=1.0, noise_std=1.0, white_back=True, all_reduce=False, warmup=2,
                 loss_coef=1.0, lambda_u=0.01, arena=None, capture_all_reduce=None, force_all_reduce=False):
        """loss_coef / lambda_u: NerfWLoss's constants for the fused loss (loss_fn=None); with a loss_fn they are its own.
        arena: the GradArena that holds the parameters' gradients (created here when None): the backward writes into it
        and the all-reduce runs on it in place.
        capture_all_reduce: record the collective INSIDE the one graph (RCCL supports stream capture); None = yes for
        the nccl backend, no otherwise (gloo cannot be captured: the step is then two graphs around an eager collective).
        force_all_reduce: issue the collective at world size 1 too (exercises RCCL on a single GPU)."""
        if not getattr(opt, "capturable", False):
            raise ValueError("GraphedTrainStep needs nerf_fl_amd.train.Adam(capturable=True)")
        import torch.distributed as dist
        self.params, self.opt, self.all_reduce = list(params), opt, bool(all_reduce)
        self.rays, self.ts, self.target = rays.detach().clone(), ts.detach().clone(), target.detach().clone()
        self.arena = arena if arena is not None else parallel.GradArena(self.params)
        self.force = bool(force_all_reduce)
        if capture_all_reduce is None:
            capture_all_reduce = self.all_reduce and dist.is_initialized() and dist.get_backend() == "nccl"
        self.captured_collective = bool(capture_all_reduce) and self.all_reduce
        dev = self.rays.device

        def fwd_bwd():
            # (parameters the backward never reaches keep the zeros the arena was created with)
            if loss_fn is None:      # NerfWLoss fused into the render kernels' per-ray epilogue (render_rays: loss_target)
                res = render_rays(models, embeddings, self.rays, self.ts, N_samples, use_disp, perturb, noise_std,
                                  N_importance, 32768, white_back, False, loss_target=self.target,
                                  loss_coef=loss_coef, lambda_u=lambda_u, grad_arena=self.arena)
                total = res["_nerfw_loss"]
            else:
                res = render_rays(models, embeddings, self.rays, self.ts, N_samples, use_disp, perturb, noise_std,
                                  N_importance, 32768, white_back, False, grad_arena=self.arena)
                total = sum(loss_fn(res, self.target).values())
            total.backward()
            key = "rgb_fine" if "rgb_fine" in res else "rgb_coarse"
            return total.detach(), psnr(res[key].detach(), self.target)

        def reduce():
            self.arena.all_reduce(force=self.force)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fwd_bwd()
                if self.all_reduce:
                    reduce()
                opt.step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        missing = [i for i, p in enumerate(self.params) if p.requires_grad and p.grad is None]
        if missing:      # a captured opt.step() would skip them for ever
            raise RuntimeError(f"GraphedTrainStep: {len(missing)} trainable parameters have no gradient after the warm-up "
                               f"steps (first: index {missing[0]}); pass only parameters the step reaches")
        self.graph = torch.cuda.CUDAGraph()
        self.graph_opt = None
        # With a process group alive, its watchdog thread polls the events of earlier collectives (hipEventQuery) at any
        # time; under the default "global" capture mode such a call from ANOTHER thread invalidates the capture
        # ("operation not permitted when stream is capturing").  thread_local confines the checks to this thread.
        mode = dict(capture_error_mode="thread_local") if dist.is_initialized() else {}
        with torch.cuda.graph(self.graph, **mode):
            self.out = fwd_bwd()
            if self.all_reduce and self.captured_collective:
                reduce()
            if not self.all_reduce or self.captured_collective:
                opt.step()
        if self.all_reduce and not self.captured_collective:
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool(), **mode):
                opt.step()

    def load(self, rays, ts, target):
        self.rays.copy_(rays)
        self.ts.copy_(ts)
        self.target.copy_(target)

    def replay(self):
        """Run the captured step on the loaded batch; returns (loss, psnr) as device scalars (valid until the next replay)."""
        self.opt.sync_hyper()
        self.graph.replay()
        if self.graph_opt is not None:
            self.arena.all_reduce(force=self.force)      # eager, in place on the arena the two graphs read and write
            self.graph_opt.replay()
        self.opt.note_replay()
        # the parameters changed behind autograd's back: move their version counters so that any eager render_rays
        # (validation) re-packs its weight streams
        torch.autograd.graph.increment_version(self.params)
        return self.out


class RayTrainer:
    """Fit coarse+fine fields to (rays, rgbs, ts) tensors that are already on the device."""

    def __init__(self, device, N_emb_xyz=10, N_emb_dir=4, N_samples=64, N_importance=64, use_disp=False,
                 perturb=1.0, noise_std=1.0, white_back=True, encode_a=False, encode_t=False, N_vocab=100,
                 N_a=48, N_tau=16, beta_min=0.1, lr=5e-4, batch_size=1024, lr_scheduler=None, num_epochs=16,
                 decay_step=(20,), decay_gamma=0.1, seed=0, use_graph=False):
        """use_graph: run the steps of fit_epoch from one captured HIP graph (GraphedTrainStep).  At the README batch of
        1024 rays the ~35 launches of an eager step are the critical path (1.9 vs 1.67 ms per step); at 4096 rays it
        makes no difference.  The first fit_epoch call spends two extra steps on its first batch (warm-up before the capture)."""
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("nerf_fl_amd.train.RayTrainer needs a ROCm device (this build has no CPU path)")
        self.use_graph = bool(use_graph)
        self._graphed = None
        self.hp = dict(N_samples=N_samples, N_importance=N_importance, use_disp=use_disp, perturb=perturb,
                       noise_std=noise_std, white_back=white_back, batch_size=batch_size)
        torch.manual_seed(seed)
        # draws of the backward's stochastic rounding: a function of the trainer's seed, and another stream on every rank
        import torch.distributed as dist
        from .rendering import set_rounding_seed
        set_rounding_seed(seed * 1000003 + (dist.get_rank() if dist.is_initialized() else 0))
        self.embeddings = {"xyz": PosEmbedding(N_emb_xyz - 1, N_emb_xyz), "dir": PosEmbedding(N_emb_dir - 1, N_emb_dir)}
        self.modules = {}                                   # checkpoint prefix -> module (train.py:48-76)
        if encode_a:
            self.embeddings["a"] = self.modules["embedding_a"] = nn.Embedding(N_vocab, N_a).to(self.dev)
        if encode_t:
            self.embeddings["t"] = self.modules["embedding_t"] = nn.Embedding(N_vocab, N_tau).to(self.dev)
        cx, cd = 6 * N_emb_xyz + 3, 6 * N_emb_dir + 3
        self.models = {"coarse": NeRF("coarse", in_channels_xyz=cx, in_channels_dir=cd).to(self.dev)}
        self.modules["nerf_coarse"] = self.models["coarse"]
        if N_importance > 0:
            self.models["fine"] = NeRF("fine", in_channels_xyz=cx, in_channels_dir=cd, encode_appearance=encode_a,
                                       in_channels_a=N_a, encode_transient=encode_t, in_channels_t=N_tau,
                                       beta_min=beta_min).to(self.dev)
            self.modules["nerf_fine"] = self.models["fine"]
        self.params = [p for m in self.modules.values() for p in m.parameters()]
        # one-launch Adam.  (torch's own fused=True variant is not an option here: it updates the parameters without
        # moving their version counters, so render_rays never re-packed its weight streams and kept rendering with
        # the initial weights -- tests/test_train_gpu.py: validation PSNR 26.89 -> 26.93 instead of 35.9)
        self.opt = Adam(self.params, lr=lr, eps=1e-8, capturable=self.use_graph)
        if lr_scheduler == "cosine":
            self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, T_max=num_epochs, eta_min=1e-8)
        elif lr_scheduler == "steplr":
            self.sched = torch.optim.lr_scheduler.MultiStepLR(self.opt, milestones=list(decay_step), gamma=decay_gamma)
        else:
            self.sched = None
        self.arena = parallel.GradArena(self.params)      # flat gradient memory: written by the backward, all-reduced in place
        self.loss = NerfWLoss()
        self.fused_loss = True          # False: the NerfWLoss module on the result dict (two extra launches), as the reference composes it
        self.gen = torch.Generator(device=self.dev).manual_seed(seed + 1)

    # ---- one optimisation step on a ready batch ------------------------------------------------
    def step(self, rays, rgbs, ts):
        hp = self.hp
        self.arena.attach()          # parameters the backward never reaches keep the zeros the arena was created with
        if self.fused_loss:
            # NerfWLoss computed in the render kernels' per-ray epilogue, its backward seeds with it
            res = render_rays(self.models, self.embeddings, rays, ts, hp["N_samples"], hp["use_disp"], hp["perturb"],
                              hp["noise_std"], hp["N_importance"], 32768, hp["white_back"], False, loss_target=rgbs,
                              loss_coef=self.loss.coef, lambda_u=self.loss.lambda_u, grad_arena=self.arena)
            total = res["_nerfw_loss"]
        else:
            res = render_rays(self.models, self.embeddings, rays, ts, hp["N_samples"], hp["use_disp"], hp["perturb"],
                              hp["noise_std"], hp["N_importance"], 32768, hp["white_back"], False, grad_arena=self.arena)
            total = sum(self.loss(res, rgbs).values())
        total.backward()
        self.arena.all_reduce()
        self.opt.step()
        key = "rgb_fine" if "rgb_fine" in res else "rgb_coarse"
        return total.detach(), psnr(res[key].detach(), rgbs)

    # ---- one epoch over device-resident data (this rank's shard) -------------------------------
    def fit_epoch(self, rays, rgbs, ts):
        n, bs = rays.shape[0], self.hp["batch_size"]
        perm = torch.randperm(n, device=self.dev, generator=self.gen)
        log = []
        for i in range(0, n - bs + 1, bs):
            idx = perm[i:i + bs]
            if self.use_graph and self.fused_loss:
                if self._graphed is None:
                    import torch.distributed as dist
                    hp = self.hp
                    self._graphed = GraphedTrainStep(
                        self.models, self.embeddings, self.params, self.opt, None, rays[idx], ts[idx], rgbs[idx],
                        hp["N_samples"], hp["N_importance"], hp["use_disp"], hp["perturb"], hp["noise_std"], hp["white_back"],
                        all_reduce=dist.is_initialized() and dist.get_world_size() > 1,
                        loss_coef=self.loss.coef, lambda_u=self.loss.lambda_u, arena=self.arena)
                self._graphed.load(rays[idx], ts[idx], rgbs[idx])
                log.append(tuple(x.clone() for x in self._graphed.replay()))      # the outputs live in the graph's pool
            else:
                log.append(self.step(rays[idx], rgbs[idx], ts[idx]))
        if self.sched is not None:
            self.sched.step()
        from .rendering import check_status
        check_status(self.dev)              # fp16 range audit of the epoch's render passes (raises FloatingPointError)
        return torch.stack([torch.stack(x) for x in log]).mean(0).tolist() if log else [math.nan, math.nan]

    @torch.no_grad()
    def validate(self, rays, rgbs, ts, chunk=32768):
        """Mean PSNR of a deterministic render (perturb 0, noise 0; train.py:176-210)."""
        hp = self.hp
        outs = []
        for i in range(0, rays.shape[0], chunk):
            res = render_rays(self.models, self.embeddings, rays[i:i + chunk], ts[i:i + chunk], hp["N_samples"],
                              hp["use_disp"], 0, 0, hp["N_importance"], chunk, hp["white_back"], False)
            outs.append(res["rgb_fine" if "rgb_fine" in res else "rgb_coarse"])
        return float(psnr(torch.cat(outs), rgbs))

    # ---- checkpoints with the reference's key prefixes -----------------------------------------
    def state_dict(self):
        return {f"{prefix}.{k}": v for prefix, m in self.modules.items() for k, v in m.state_dict().items()}

    def load_state_dict(self, sd):
        """Accepts this harness' checkpoints and Lightning checkpoints' `state_dict` (prefix-stripped per
        module exactly as utils.extract_model_state_dict does)."""
        sd = sd.get("state_dict", sd)
        for prefix, m in self.modules.items():
            sub = {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}
            m.load_state_dict(sub, strict=True)

    def save(self, path, epoch=0):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save({"epoch": epoch, "state_dict": self.state_dict()}, path)

    def load(self, path):
        self.load_state_dict(torch.load(path, map_location=self.dev, weights_only=True))
