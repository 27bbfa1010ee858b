#!/usr/bin/env python3
"""bench.py -- ray-samples/sec of the render_rays hot path on N MI355X.

Workload (BASELINE.json configs[1]): Blender-lego-like rays (near 2, far 6),
batch_size = 4096 rays per GPU, N_samples = 64 + N_importance = 64, base NeRF
coarse + fine, training-mode sampling (perturb = 1, noise_std = 1, white
background), synthetic rays/targets and seeded random-init weights, batch already
resident in HBM.

  --mode train  (default) one step = render_rays forward + colour loss (losses.py:35-41)
                + hand-written backward + gradient all-reduce (N > 1) + Adam step
                (lr 5e-4, eps 1e-8: utils/__init__.py:30-32) + weight re-pack
  --mode render one step = render_rays forward only (pack + coarse + sample_pdf + fine)
  --workload cfg3   configs[3] per-GPU shape instead (Phototourism: NeRF-W a+t, N_vocab 1500, black background,
                    per-ray near/far; --rays 1024 is the README batch)

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|render]

N > 1: run plainly (`python bench.py --gpus N`), this process then starts N ranks itself -- one fresh process per
GPU, rendezvous on 127.0.0.1, and exits non-zero if any rank fails -- or run under a launcher
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`), which is detected by WORLD_SIZE.
Rays shard across ranks with no data-path collective (weak scaling: every rank renders its own batch);
training adds ONE flat RCCL all-reduce of the gradients per step.  Rank 0 prints ONE JSON line; `n_gpus` in
it is the size of the process group that actually ran.

  --dry   rehearsal of the multi-rank plumbing without a GPU: gloo, CPU tensors, the same launcher, the same flat
          gradient all-reduce over stand-in parameters (tests/test_bench_launcher.py)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SAMPLES, N_IMPORTANCE = 64, 64
# algorithmic FLOPs (2 x weight-matrix MACs) per field evaluation, SURVEY.md section 8(d)
FLOP_EVAL = {"base": 2 * 593408, "at": 2 * 684160}
FOLDED_FLOP_EVAL = 2 * 256 * 256     # xyz_encoding_final: no tiles in the packed streams
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
def _latest_summary():
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_summary.json")))
    return hits[-1] if hits else None


PROFILE_SUMMARY = _latest_summary()     # the PMC passes (rocprofv3 --pmc, separate runs) cannot be made inside this process


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="train", choices=["train", "render"])
    ap.add_argument("--workload", default="cfg1", choices=["cfg1", "cfg3"])
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU and step")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f16"])
    ap.add_argument("--graph", action="store_true", help="train mode: replay the step from one captured HIP graph")
    ap.add_argument("--unfused-loss", action="store_true", help="NerfWLoss as a module on the result dict (two more launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the warm-up and the timed steps (no render-only companion, roofline launches or CPU baseline): "
                         "the command the PMC passes of profiles/collect.sh count bytes over")
    ap.add_argument("--render-steps", type=int, default=0,
                    help="with --no-extras: this many forward-only steps after the timed ones (gives the PMC passes the "
                         "inference instantiation of the render kernel to count bytes on)")
    ap.add_argument("--dry", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and issue the gradient all-reduce at world size 1 too: exercises the "
                         "RCCL path (init, in-place all-reduce on the gradient arena, capture in the step's graph) on one GPU")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="after the timed steps: keep stepping for this long with one HIP-event pair per step and report "
                         "median / p90 (`sustained`); 0 = off")
    ap.add_argument("--only-default-backward", action="store_true",
                    help="skip the companion measurements of the other backward arithmetics (the rocprofv3 --stats pass of "
                         "profiles/collect.sh: keeps one population per kernel in the trace)")
    ap.add_argument("--backward", default="f16", choices=["f16", "f16w", "f16x3"],
                    help="arithmetic of the MLP backward: f16 = single fp16 product (default), f16w = the gradient chain reads "
                         "hi+lo weight fragments (two products: no systematic training-curve offset), f16x3 = split operands, "
                         "3 products, hi+lo stashes (fp32-class, the reference's precision class)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher (parent process; never touches the GPU)
# ------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    """Start args.gpus fresh ranks of this script, one per device; return the worst exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   NFL_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stderr=subprocess.PIPE,
                                      text=True, bufsize=1))
    import threading

    def pump(r, stream):            # every line a rank writes to stderr is passed on with its rank in front
        for line in stream:
            sys.stderr.write(f"[rank {r}] {line}")
        stream.close()

    pumps = [threading.Thread(target=pump, args=(r, p.stderr), daemon=True) for r, p in enumerate(procs)]
    for t in pumps:
        t.start()
    rc, deadline = 0, None
    pending = set(range(args.gpus))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0:
                rc = rc or code
                print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
                if deadline is None:
                    deadline = time.time() + 20        # the others are stuck in a collective: do not wait for ever
        if deadline is not None and time.time() > deadline:
            for r in pending:
                procs[r].kill()                          # exact PIDs this process started
        time.sleep(0.05)
    for t in pumps:
        t.join(timeout=5)
    return rc


# ------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = CPU restatement of the reference, eager PyTorch fp32, host cores)
# ------------------------------------------------------------------------------------------------
def cpu_baseline(mode, n_rays=4096):
    """Forward (best of 3) and, in train mode, one full train step (forward + NerfWLoss + backward through autograd,
    1 repetition) of configs[1] on the host cores.  A 1-GPU box gives this job a 16-CPU share."""
    import torch
    from oracle import nerfw_oracle as orc
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.manual_seed(0)
    spec_c, spec_f = orc.FieldSpec("coarse"), orc.FieldSpec("fine")
    P_c, P_f = orc.make_field_params(spec_c, 11, "sharp"), orc.make_field_params(spec_f, 12, "sharp")
    rays = orc.make_rays(n_rays, 5)
    kw = dict(n_samples=N_SAMPLES, n_importance=N_IMPORTANCE, perturb=1.0, noise_std=1.0, white_back=True,
              perturb_rand=torch.rand(n_rays, N_SAMPLES), noise_coarse=torch.randn(n_rays, N_SAMPLES),
              u=torch.rand(n_rays, N_IMPORTANCE), noise_fine=torch.randn(n_rays, N_SAMPLES + N_IMPORTANCE))
    units = n_rays * (N_SAMPLES + N_IMPORTANCE)
    best = float("inf")
    with torch.no_grad():
        for _ in range(3):
            t0 = time.perf_counter()
            orc.render_rays(spec_c, P_c, spec_f, P_f, rays, **kw)
            best = min(best, time.perf_counter() - t0)
    out = {"value": units / best, "unit": "ray-samples/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n_rays} rays x (64+64), base NeRF coarse+fine, forward render_rays, best of 3 "
                     f"({best:.2f} s each); oracle/nerfw_oracle.py"}
    if mode == "train":
        for P in (P_c, P_f):
            for p in P.values():
                p.requires_grad_(True)
        target = torch.rand(n_rays, 3)
        t0 = time.perf_counter()
        res = orc.render_rays(spec_c, P_c, spec_f, P_f, rays, **kw)
        sum(orc.nerfw_loss(res, target).values()).backward()
        dt = time.perf_counter() - t0
        out = {"value": units / dt, "unit": "ray-samples/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{n_rays} rays x (64+64), base NeRF coarse+fine, one train step = forward render_rays + "
                         f"NerfWLoss + autograd backward (no optimizer), 1 repetition ({dt:.2f} s); "
                         "oracle/nerfw_oracle.py",
               "forward_value": out["value"], "forward_sample": out["sample"]}
    return out


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
def build_models(dev, workload):
    import torch
    from nerf_fl_amd import NeRF, PosEmbedding, synth
    at = workload == "cfg3"
    models = {}
    for typ, seed in (("coarse", 11), ("fine", 12)):
        fine_kw = dict(encode_appearance=at, encode_transient=at) if typ == "fine" else {}
        m = NeRF(typ, beta_min=0.03, **fine_kw)
        m.load_state_dict(synth.make_field_params(seed, "sharp", typ=typ, **fine_kw))
        models[typ] = m.to(dev)
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    if at:
        torch.manual_seed(5)
        emb["a"] = torch.nn.Embedding(1500, 48).to(dev)
        emb["t"] = torch.nn.Embedding(1500, 16).to(dev)
    return models, emb


def run_dry(args, rank, world):
    """gloo rehearsal: process group, the flat gradient all-reduce over stand-in parameters, barrier + MAX timing."""
    import torch
    import torch.distributed as dist
    from nerf_fl_amd import parallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("NFL_BENCH_FAIL_RANK") == str(rank):
        os._exit(3)                                            # failure injection for the launcher test
    torch.manual_seed(rank)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (595844, 595844, 1500 * 64)]
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for p in params:
            p.grad = torch.full_like(p, float(rank + 1))
        parallel.all_reduce_gradients(params)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    expect = sum(range(1, world + 1)) / world
    ok = all(bool((p.grad == expect).all()) for p in params)
    if rank == 0:
        print(json.dumps({"metric": "ray-samples/sec (64+64)", "value": None, "unit": "ray-samples/s",
                          "n_gpus": dist.get_world_size(), "ranks": dist.get_world_size(), "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * float(t.item()) / max(args.steps, 1), "dry": True,
                          "backend": "gloo", "allreduce_ok": ok, "data": "synthetic"}))
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 4


def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has WORLD_SIZE={world}; "
                         "refusing to report a number for a different GPU count than asked")
    if args.dry:
        return run_dry(args, rank, world)

    import torch
    dist = None
    backend = None
    if world > 1 or args.force_dist:
        import datetime

        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        backend = os.environ.get("NFL_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of the N>1 path on one GPU
        if os.environ.get("NFL_BENCH_ONE_DEVICE"):
            local_rank = 0
        rdv = datetime.timedelta(seconds=float(os.environ.get("NFL_BENCH_RDV_TIMEOUT", "180")))   # a rank that never arrives fails the run
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=rdv, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=rdv)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import nerf_fl_amd
    from nerf_fl_amd import parallel, render_rays, synth
    from nerf_fl_amd import rendering as rnd
    from nerf_fl_amd.train import Adam, GraphedTrainStep, NerfWLoss
    nerf_fl_amd.set_precision(args.precision, backward=args.backward)

    R = args.rays
    cfg3 = args.workload == "cfg3"
    white_back = not cfg3
    models, emb = build_models(dev, args.workload)
    torch.manual_seed(1234 + rank)
    rays = (synth.make_rays_photo if cfg3 else synth.make_rays)(R, 100 + rank).to(dev)   # each rank: its own shard
    ts = torch.randint(0, 1500, (R,), device=dev) if cfg3 else torch.zeros(R, dtype=torch.long, device=dev)
    target = torch.rand(R, 3, device=dev)
    modules = list(models.values()) + [emb[k] for k in ("a", "t") if k in emb]
    params = [p for m in modules for p in m.parameters()]
    loss_fn = NerfWLoss()
    opt = Adam(params, lr=5e-4, eps=1e-8, capturable=args.graph)
    arena = parallel.GradArena(params)       # flat gradient memory: written by the HIP backward, all-reduced in place
    force = bool(args.force_dist)

    def render_step():
        with torch.no_grad():
            return render_rays(models, emb, rays, ts, N_SAMPLES, False, 1.0, 1.0, N_IMPORTANCE, 32768, white_back, False)

    def train_step():
        if args.unfused_loss:
            res = render_rays(models, emb, rays, ts, N_SAMPLES, False, 1.0, 1.0, N_IMPORTANCE, 32768, white_back, False,
                              grad_arena=arena)
            loss = sum(loss_fn(res, target).values())  # c_l + f_l (+ b_l + s_l), forward and backward one launch each
        else:       # NerfWLoss and its backward seeds in the render kernels' per-ray epilogue (losses.py:35-50)
            loss = render_rays(models, emb, rays, ts, N_SAMPLES, False, 1.0, 1.0, N_IMPORTANCE, 32768, white_back, False,
                               loss_target=target, grad_arena=arena)["_nerfw_loss"]
        loss.backward()
        if dist is not None:
            arena.all_reduce(force=force)       # one in-place collective on the memory the backward wrote
        opt.step()

    graphed = None
    if args.mode == "train" and args.graph:
        graphed = GraphedTrainStep(models, emb, params, opt, loss_fn if args.unfused_loss else None, rays, ts, target,
                                   N_SAMPLES, N_IMPORTANCE, white_back=white_back, all_reduce=dist is not None,
                                   arena=arena, force_all_reduce=force)
        step = graphed.replay
    else:
        step = train_step if args.mode == "train" else render_step

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_ranks = dist.get_world_size() if dist is not None else 1
    # sustained rate (>= 100 timed iterations, event-timed, median): outside the driver-specified steps, the
    # same step for >= --sustained-seconds with one HIP event per step on the launch stream.  The step count is fixed
    # from the (rank-maximised) timing above, so that every rank issues the same number of collectives.
    sustained = None
    if args.sustained_seconds > 0:
        n_sus = max(100, int(args.sustained_seconds / max(elapsed / args.steps, 1e-6)) + 1)
        sustained = event_timed(step, n_sus)
        sync()
    # The same step with the other two backward arithmetics, measured on the same batch right after the default one (same
    # K / W).  The reference differentiates in fp32 (train.py:158-174); the default backward here multiplies in fp16.
    #   f16w : the gradient chain reads hi + lo weight fragments -- no systematic training-curve offset
    #   f16x3: operands split everywhere, three products, split stashes -- the train-step figure at the reference's
    #          precision class (`value_fp32_class`)
    other_modes = {}
    if (args.mode == "train" and args.precision == "f16x3" and args.backward == "f16" and not args.no_extras
            and not args.only_default_backward):
        descr = {"f16w": "as the default (fp16 MFMA, 1 product, fp16 stashes, loss scale) but the gradient chain W^T delta reads "
                         "hi+lo weight fragments (2 products): weights to fp32 class in the chain",
                 "f16x3": "fp16 MFMA, weights / activations / gradients split hi+lo, 3 products, fp32 accumulate; "
                          "hi+lo activation and gradient stashes (2x the bytes); gradients returned in fp32"}
        for mode in ("f16w", "f16x3"):
            nerf_fl_amd.set_precision(backward=mode)
            stepm = train_step
            if args.graph:
                gm = GraphedTrainStep(models, emb, params, opt, loss_fn if args.unfused_loss else None, rays, ts, target,
                                      N_SAMPLES, N_IMPORTANCE, white_back=white_back, all_reduce=dist is not None,
                                      arena=arena, force_all_reduce=force)
                stepm = gm.replay
            for _ in range(args.warmup):
                stepm()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                stepm()
            sync()
            em = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([em], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                em = float(t.item())
            other_modes[mode] = {"value": R * (N_SAMPLES + N_IMPORTANCE) * n_ranks * args.steps / em, "ms_per_step": 1e3 * em / args.steps,
                                 "dtype": f"f16x3 fwd + {mode} bwd", "backward_arithmetic": descr[mode]}
            if args.sustained_seconds > 0:
                other_modes[mode]["sustained"] = event_timed(stepm, max(100, int(args.sustained_seconds / max(em / args.steps, 1e-6)) + 1))
                sync()
        nerf_fl_amd.set_precision(backward="f16")
    fp32_class = other_modes.get("f16x3")
    sync_diff = None
    if dist is not None:
        # data parallelism keeps the replicas identical: same seeded weights, the same averaged gradients, the same Adam.
        # Largest difference between this rank's parameters and rank 0's, maximised over ranks: must be exactly 0.
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        d = (flat - ref).abs().max().reshape(1)
        dist.all_reduce(d, op=dist.ReduceOp.MAX)
        sync_diff = float(d.item())
    F = N_SAMPLES + N_IMPORTANCE
    total_units = R * F * n_ranks * args.steps
    fine_kind = "at" if cfg3 else "base"
    wl = ("configs[3] per-GPU shape: phototourism-like rays (per-ray near/far), NeRF-W a+t fine field, N_vocab 1500, "
          "black background" if cfg3 else
          "configs[1]: lego-like rays, base NeRF coarse+fine, white_back")
    out = {
        "metric": "ray-samples/sec (64+64)",
        "value": total_units / elapsed,
        "unit": "ray-samples/s",
        "n_gpus": n_ranks,
        "ranks": n_ranks,
        "backend": backend,
        "all_reduce": (None if dist is None or args.mode != "train" else
                       {"tensor": "one flat fp32 gradient arena, in place", "numel": int(arena.flat.numel()),
                        "forced_at_world_1": force and n_ranks == 1,
                        "captured_in_graph": bool(graphed is not None and graphed.captured_collective)}),
        "replica_param_max_diff": sync_diff,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "sustained": sustained,
        "value_fp32_class": None if fp32_class is None else fp32_class["value"],
        "fp32_class": fp32_class,
        "value_exact_weight_chain": None if "f16w" not in other_modes else other_modes["f16w"]["value"],
        "exact_weight_chain": other_modes.get("f16w"),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": (f"f16x3 fwd + {args.backward} bwd" if args.mode == "train" else "f16x3") if args.precision == "f16x3" else "f16",
        "data": "synthetic",
        "config": {"workload": f"{wl}; {R} rays/GPU, N_samples=64 + N_importance=64, perturb=1 noise_std=1; "
                               + ("train step: render_rays fwd + NerfWLoss + HIP backward + grad all-reduce + Adam + re-pack"
                                  + (", replayed from one HIP graph" if args.graph else "")
                                  if args.mode == "train" else
                                  "forward render_rays (pack + coarse pass + sample_pdf + fine pass)"),
                   "rays_per_gpu": R, "mode": args.mode, "precision": args.precision,
                   "forward_arithmetic": "fp16 MFMA, operands split hi+lo, 3 products, fp32 accumulate (f16x3)"
                                         if args.precision == "f16x3" else "fp16 MFMA, 1 product, fp32 accumulate",
                   "backward_arithmetic": ("fp16 MFMA, 1 product, fp32 accumulate, fp16 activation/gradient stashes "
                                           "under a device-chosen power-of-two loss scale, gradients and chain weights rounded to fp16 "
                                           "stochastically (zero-mean); gradients returned in fp32"
                                           if args.backward == "f16" else
                                           "as f16, the gradient chain reading hi+lo weight fragments (2 products)" if args.backward == "f16w" else
                                           "fp16 MFMA, operands split hi+lo, 3 products, fp32 accumulate, hi+lo stashes (f16x3)"),
                   "mlp_evals_per_ray": N_SAMPLES + F, "hip_graph": bool(args.graph)},
    }

    if args.no_extras:
        for _ in range(args.render_steps):
            render_step()
        sync()
    if args.mode == "train" and not args.no_extras:
        # forward-only throughput of the same batch, reported beside the train-step value
        for _ in range(3):
            render_step()
        sync()
        t0 = time.perf_counter()
        for _ in range(20):
            render_step()
        sync()
        out["render_only_value"] = R * F * n_ranks * 20 / (time.perf_counter() - t0)

    if rank == 0:
        if not args.no_extras:
            out.update(roofline(args, rnd, models, emb, rays, ts, dev, fine_kind, white_back))
        if n_ranks == 1 and not args.no_cpu_baseline and not args.no_extras:
            out["cpu_baseline"] = cpu_baseline(args.mode)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def event_timed(fn, n):
    """Run fn() n times with one HIP event after each call on the current stream; per-call durations (the time from
    the previous event: kernels and any host gap between them) -> median / p90 / mean in ms."""
    import torch
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    evs[0].record()
    for i in range(n):
        fn()
        evs[i + 1].record()
        if i % 64 == 63:
            evs[i - 31].synchronize()        # keep the host at most a few dozen steps ahead of the device
    torch.cuda.synchronize()
    d = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
    return {"median_ms": d[n // 2], "p90_ms": d[min(n - 1, int(0.9 * n))], "mean_ms": sum(d) / n, "min_ms": d[0], "steps": n}


def roofline(args, rnd, models, emb, rays, ts, dev, fine_kind, white_back):
    """Roofline of the dominant kernel: the fine-pass launch of nfl_render_kernel (R rays x 128 samples), timed
    alone with events on the launch stream, in the instantiation the timed mode runs (MODE 1 = training forward,
    writes the activation stash; MODE 0 = inference) and, beside it, in the other one."""
    import torch
    from nerf_fl_amd import _lib
    R = rays.shape[0]
    F = N_SAMPLES + N_IMPORTANCE
    f_f = rnd._field(models["fine"], 10, 4, dev)
    z = torch.sort(rays[:, 6:7] + (rays[:, 7:8] - rays[:, 6:7]) * torch.rand(R, F, device=dev), dim=1)[0]
    noise = torch.randn(R, F, device=dev)
    lat = {}
    if fine_kind == "at":
        lat = dict(a_emb=emb["a"](ts).detach(), t_emb=emb["t"](ts).detach())
    flops = R * F * FLOP_EVAL[fine_kind]

    def time_pass(stash):
        run = lambda: rnd._run_pass(f_f, rays, F, z=z, noise=None if lat else noise, noise_std=1.0,
                                    white_back=white_back, stash=stash, **lat)
        for _ in range(5):
            run()
        # >= 100 launches, one event pair each, on the stream the kernel is launched on (torch's current stream)
        return event_timed(run, 200 if args.sustained_seconds > 0 else 30)

    base = _lib.lib().nfl_render_kernel_name(rnd._PREC[args.precision], 10).decode()      # "...<3, 1, 10, 0>"
    names = {False: base, True: base[:-2] + "1>"}
    # HBM bytes per launch from the committed PMC profile (FETCH_SIZE doubled as the gfx950 note in
    # MI355X_MICROARCH.md prescribes, + WRITE_SIZE); collected by rocprofv3 --pmc in separate passes, so it cannot be
    # measured inside this process
    summ = {}
    try:
        summ = json.load(open(PROFILE_SUMMARY))
        # where the byte counts come from, and whether the library they were counted on is the one running now
        import hashlib
        prov = dict(summ.get("provenance") or {})
        prov["file"] = os.path.relpath(PROFILE_SUMMARY, ROOT)
        cur = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]
        prov["matches_running_library"] = (prov.get("lib_sha16") == cur) if prov.get("lib_sha16") else None
        summ["provenance"] = prov
    except Exception:
        pass

    def traffic_of(kname):
        for k, v in summ.get("traffic", {}).items():
            if kname.split("(")[0] in k:
                return v.get("hbm_bytes_max_launch")
        return None

    def issued_of(kname, ms):
        """MFMA work the kernel ISSUES per launch (PMC SQ_INSTS_MFMA of the committed profile x 32*32*16*2 FLOP per
        v_mfma_f32_32x32x16_f16) over the launch time measured here: the figure to hold against what the chip sustains on
        random data (MI355X_MICROARCH.md, 'DVFS give-back': 1,247 TFLOP/s at 1.90-1.95 GHz for a dense MFMA loop), which is
        what bounds an f16x3 kernel -- three issued products per algorithmic one."""
        for k, v in summ.get("sq", {}).items():
            if kname.split("(")[0] in k and v.get("SQ_INSTS_MFMA"):
                n = float(v["SQ_INSTS_MFMA"])
                if R * F != 4096 * 128 or fine_kind != "base":          # the profile counts the default workload's fine-pass launch
                    return None
                tf = n * 32768.0 / (ms * 1e-3) / 1e12
                return {"mfma_insts_per_launch": n, "flop_per_inst": 32768, "tflops": tf,
                        "sustained_dense_mfma_random_data_tflops": 1247.0, "frac_of_sustained": tf / 1247.0,
                        "pipe_busy_frac_pmc": v.get("mfma_busy_frac")}
        return None

    def entry(stash):
        t = time_pass(stash)
        ms = t["mean_ms"]                 # average launch duration over the timed region (what rocprofv3 --stats averages too)
        ach = flops / (ms * 1e-3) / 1e12
        executed = FLOP_EVAL[fine_kind] - FOLDED_FLOP_EVAL
        return {"bound": "mfma", "achieved": ach, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "issued": issued_of(names[stash], ms),
                "frac": ach / PEAK_F16_MFMA_TFLOPS, "traffic": traffic_of(names[stash]), "kernel": names[stash],
                "launch_ms": ms, "launch_sustained": t, "traffic_source": summ.get("provenance"),
                "executed_flops_per_eval": executed,
                "frac_of_executed_flops": ach / PEAK_F16_MFMA_TFLOPS * executed / FLOP_EVAL[fine_kind],
                "note": f"algorithmic FLOPs of the reference's field ({FLOP_EVAL[fine_kind]} per evaluation x {R} rays x {F} samples); "
                        f"the kernel executes {executed} of them (the linear xyz_encoding_final, 2 x 256 x 256, is folded into the "
                        "layers that read it) and f16x3 issues 3 MFMA products per executed product"}

    train = args.mode == "train" and args.precision == "f16x3"
    out = {"roofline": entry(train)}
    if args.precision == "f16x3":
        out["roofline_inference" if train else "roofline_training"] = entry(not train)
    st = summ.get("step_traffic")
    if st and args.mode == "train":
        st = dict(st, source=summ.get("provenance"))
        out["step_traffic"] = st       # PMC HBM bytes of one whole train step vs its algorithmic bytes (profiles/)
    return out


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
