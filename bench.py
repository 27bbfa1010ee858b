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

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|render]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rays shard across ranks with no data-path collective (weak scaling: every rank
renders its own 4096-ray batch); training adds ONE flat RCCL all-reduce of the
gradients per step.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_PER_GPU = 4096
N_SAMPLES, N_IMPORTANCE = 64, 64
# algorithmic FLOPs (2 x weight-matrix MACs) per field evaluation, SURVEY.md section 8(d)
FLOP_BASE_EVAL = 2 * 593408
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md


def build_models(dev):
    from nerf_fl_amd import NeRF, PosEmbedding
    from oracle import nerfw_oracle as orc     # only for the seeded weights / rays (shared with the tests)
    spec = orc.FieldSpec("coarse")
    models = {}
    for typ, seed in (("coarse", 11), ("fine", 12)):
        m = NeRF(typ)
        m.load_state_dict(orc.make_field_params(orc.FieldSpec(typ), seed, "sharp"))
        models[typ] = m.to(dev)
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    return models, emb, spec


def cpu_baseline(n_rays=4096, reps=3):
    """The oracle (CPU restatement of the reference, eager PyTorch fp32) on the host cores.
    A 1-GPU box gives this job a 16-CPU share; more threads than that only oversubscribe."""
    from oracle import nerfw_oracle as orc
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.manual_seed(0)
    spec_c, spec_f = orc.FieldSpec("coarse"), orc.FieldSpec("fine")
    P_c, P_f = orc.make_field_params(spec_c, 11, "sharp"), orc.make_field_params(spec_f, 12, "sharp")
    rays = orc.make_rays(n_rays, 5)
    kw = dict(n_samples=N_SAMPLES, n_importance=N_IMPORTANCE, perturb=1.0, noise_std=1.0, white_back=True,
              perturb_rand=torch.rand(n_rays, N_SAMPLES), noise_coarse=torch.randn(n_rays, N_SAMPLES),
              u=torch.rand(n_rays, N_IMPORTANCE), noise_fine=torch.randn(n_rays, N_SAMPLES + N_IMPORTANCE))
    best = float("inf")
    with torch.no_grad():
        for _ in range(reps):
            t0 = time.perf_counter()
            orc.render_rays(spec_c, P_c, spec_f, P_f, rays, **kw)
            best = min(best, time.perf_counter() - t0)
    return {"value": n_rays * (N_SAMPLES + N_IMPORTANCE) / best, "unit": "ray-samples/s",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_rays} rays x (64+64), base NeRF coarse+fine, forward render_rays, best of {reps} "
                      f"({best:.2f} s each); oracle/nerfw_oracle.py"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="train", choices=["train", "render"])
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("NFL_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of the N>1 path on one GPU
        if os.environ.get("NFL_BENCH_ONE_DEVICE"):
            local_rank = 0
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import nerf_fl_amd
    from nerf_fl_amd import render_rays
    from nerf_fl_amd import rendering as rnd
    from oracle import nerfw_oracle as orc
    nerf_fl_amd.set_precision(args.precision)

    models, emb, _ = build_models(dev)
    torch.manual_seed(1234 + rank)
    rays = orc.make_rays(R_PER_GPU, 100 + rank).to(dev)       # each rank: its own shard of rays
    ts = torch.zeros(R_PER_GPU, dtype=torch.long, device=dev)

    target = torch.rand(R_PER_GPU, 3, device=dev)
    params = [p for m in models.values() for p in m.parameters()]
    from nerf_fl_amd.train import Adam, NerfWLoss   # torch.optim.Adam's arithmetic in one launch (C ABI nfl_adam_step)
    loss_fn = NerfWLoss()
    opt = Adam(params, lr=5e-4, eps=1e-8)
    from nerf_fl_amd import parallel

    def render_step():
        with torch.no_grad():
            return render_rays(models, emb, rays, ts, N_SAMPLES, False, 1.0, 1.0, N_IMPORTANCE, 32768, True, False)

    def train_step():
        opt.zero_grad(set_to_none=True)
        res = render_rays(models, emb, rays, ts, N_SAMPLES, False, 1.0, 1.0, N_IMPORTANCE, 32768, True, False)
        loss = sum(loss_fn(res, target).values())      # c_l + f_l (losses.py:35-41), forward and backward one launch each
        loss.backward()
        if dist is not None:
            parallel.all_reduce_gradients(params)
        opt.step()

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

    total_units = R_PER_GPU * (N_SAMPLES + N_IMPORTANCE) * world * args.steps
    out = {
        "metric": "ray-samples/sec (64+64)",
        "value": total_units / elapsed,
        "unit": "ray-samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16x3" if args.precision == "f16x3" else "f16",
        "data": "synthetic",
        "config": {"workload": "configs[1]: lego-like rays 4096/GPU, N_samples=64 + N_importance=64, base NeRF "
                               "coarse+fine, perturb=1 noise_std=1 white_back; "
                               + ("train step: render_rays fwd + colour loss + HIP backward + grad all-reduce + Adam"
                                  if args.mode == "train" else
                                  "forward render_rays (pack + coarse pass + sample_pdf + fine pass)"),
                   "rays_per_gpu": R_PER_GPU, "mode": args.mode, "precision": args.precision,
                   "mlp_evals_per_ray": N_SAMPLES + N_SAMPLES + N_IMPORTANCE},
    }

    if args.mode == "train":
        # forward-only throughput of the same batch, reported beside the train-step value
        for _ in range(3):
            render_step()
        sync()
        t0 = time.perf_counter()
        for _ in range(20):
            render_step()
        sync()
        out["render_only_value"] = R_PER_GPU * (N_SAMPLES + N_IMPORTANCE) * world * 20 / (time.perf_counter() - t0)

    if rank == 0:
        # ---- roofline of the dominant kernel: the fine-pass launch of nfl_render_kernel
        # (4096 rays x 128 samples x 1.1868 MFLOP), timed alone with events on the launch stream
        f_f = rnd._field(models["fine"], 10, 4, dev)
        F = N_SAMPLES + N_IMPORTANCE
        z = torch.sort(2 + 4 * torch.rand(R_PER_GPU, F, device=dev), dim=1)[0]
        noise = torch.randn(R_PER_GPU, F, device=dev)
        for _ in range(5):
            rnd._run_pass(f_f, rays, F, z=z, noise=noise, noise_std=1.0, white_back=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            rnd._run_pass(f_f, rays, F, z=z, noise=noise, noise_std=1.0, white_back=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = R_PER_GPU * F * FLOP_BASE_EVAL
        achieved = flops / (ms * 1e-3) / 1e12
        from nerf_fl_amd import _lib
        kname = _lib.lib().nfl_render_kernel_name(rnd._PREC[args.precision], 10).decode()
        # HBM bytes per launch of this kernel from the committed PMC profile (FETCH_SIZE doubled as the
        # gfx950 note in MI355X_MICROARCH.md prescribes, + WRITE_SIZE); collected by rocprofv3 --pmc in
        # separate passes, so it cannot be measured inside this process
        traffic = None
        try:
            summ = json.load(open(os.path.join(ROOT, "profiles", "r01g_summary.json")))
            for k, v in summ["traffic"].items():
                if kname.split("(")[0] in k:
                    traffic = v["hbm_bytes_max_launch"]
        except Exception:
            pass
        out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / PEAK_F16_MFMA_TFLOPS, "traffic": traffic,
                           "kernel": kname, "launch_ms": ms,
                           "note": "algorithmic FLOPs (2 x 593408 MAC per field evaluation x 4096 rays x 128 samples); "
                                   "f16x3 issues 3 MFMA products per algorithmic product"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
