"""Sweep of the forward's per-layer product plan (csrc/nfl_prods.h; not a test).  For every variant library
nerf_fl_amd/libnerf_fl_amd_p<tag>.so (built by `python tests/report_prods.py --build`, which compiles the x3 render TU
with one layer group's correction product dropped) this prints the worst output error over ALL two-pass fixtures with
the reference's fine depths injected (what test_render_at_reference_depths holds to 1e-4) and the fine-pass launch time,
so that each layer's contribution to the error and its price in time can be read off.
    python tests/report_prods.py --build [-j N]        (build container; ~1 min per variant)
    python tests/report_prods.py                       (GPU box)
    python tests/report_prods.py --one <lib or ''>     (worker: one library, JSON line)"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "nerf_fl_amd", "csrc")
IDX = ["L1", "L2", "L3", "L4", "L5", "L6", "L7", "L8", "SIG", "DIR", "RGB", "T1", "T2", "T3", "T4", "THEAD"]
GROUPS = {"L1": ["L1"], "L2": ["L2"], "L3": ["L3"], "L4": ["L4"], "L5": ["L5"], "L6": ["L6"], "L7": ["L7"], "L8": ["L8"],
          "SIG": ["SIG"], "DIR": ["DIR"], "RGB": ["RGB"], "TR": ["T1", "T2", "T3", "T4", "THEAD"]}


def plan_of(tag):
    """tag = '<group>w' (drop w_lo x_hi) | '<group>x' (drop w_hi x_lo) | 'plan_<16 digits>'"""
    if tag.startswith("plan_"):
        return [int(c) for c in tag[5:]]
    grp, kind = tag[:-1], tag[-1]
    plan = [3] * 16
    for name in GROUPS[grp]:
        plan[IDX.index(name)] = 2 if kind == "w" else 1
    return plan


def build(tags, jobs):
    procs = []
    for tag in tags:
        flags = "-DNFL_DIAG_INFERENCE_ONLY '-DNFL_PRODS_OVERRIDE={" + ",".join(map(str, plan_of(tag))) + "}'"
        cmd = f"make -s -C {CSRC} variant VTU=nfl_render_x3 VNAME=_p{tag} VFLAGS=\"{flags}\""
        procs.append((tag, subprocess.Popen(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        while sum(p.poll() is None for _, p in procs) >= jobs:
            import time
            time.sleep(1)
    for tag, p in procs:
        out, _ = p.communicate()
        print(tag, "ok" if p.returncode == 0 else "FAILED\n" + out[-2000:], flush=True)


def one():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch

    import golden_util as gu
    import gpu_util
    import test_parity_gpu as T
    from nerf_fl_amd import rendering as rnd
    worst, per = (0.0, ""), {}
    for n in T.RENDER:
        cfg, a = gu.load(n)
        if cfg["I"] == 0:
            continue
        specs, kw = gu.oracle_kwargs(cfg, a)
        kw["z_fine"] = a["z_fine"]
        if n.startswith("g18"):          # other encoder widths: not in the sweep libraries (built before they existed)
            continue
        got = gpu_util.hip_render(specs, a["rays"], kw, precision="f16x3")
        errs = {k: (got[k] - a["out." + k]).abs().max().item() for k in cfg["keys"]}
        k = max(errs, key=errs.get)
        per[n] = errs[k]
        worst = max(worst, (errs[k], f"{n}:{k}"))
    # fine-pass launch time, inference instantiation, 4096 rays x 128 samples (bench.py's roofline launch)
    from oracle import nerfw_oracle as orc
    dev = torch.device("cuda", 0)
    spec = orc.FieldSpec("fine")
    m = gpu_util.module_from(spec, orc.make_field_params(spec, 12, "sharp"))
    f = rnd._field(m, 10, 4, dev)
    rays = orc.make_rays(4096, 100).to(dev)
    z = torch.sort(2 + 4 * torch.rand(4096, 128, device=dev), dim=1)[0]
    noise = torch.randn(4096, 128, device=dev)
    run = lambda: rnd._run_pass(f, rays, 128, z=z, noise=noise, noise_std=1.0, white_back=True)
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(40):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"worst": worst[0], "where": worst[1], "fine_ms": e0.elapsed_time(e1) / 40,
                      "trained": max(v for k, v in per.items() if k.startswith("g17")),
                      "xyz15": max([v for k, v in per.items() if "xyz15" in k] or [0.0])}))


def main():
    if "--one" in sys.argv:
        return one()
    tags = [g + k for g in GROUPS for k in "wx"]
    extra = [a for a in sys.argv[1:] if a.startswith("plan_")]
    if "--build" in sys.argv:
        jobs = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else 4
        return build(extra or tags, jobs)
    import glob
    libs = [""] + sorted(glob.glob(os.path.join(ROOT, "nerf_fl_amd", "libnerf_fl_amd_p*.so")))
    print(f"{'variant':28s} {'worst err':>10s} {'trained':>10s} {'xyz15':>10s} {'fine ms':>8s}  where")
    for lib in libs + [""]:
        env = dict(os.environ, NERF_FL_AMD_DEV="1")
        if lib:
            env["NFL_LIB"] = lib
        else:
            env.pop("NFL_LIB", None)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True)
        name = os.path.basename(lib)[len("libnerf_fl_amd_p"):-3] if lib else "mainline (3 everywhere)"
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(f"{name:28s} ERROR {p.stderr[-300:]}", flush=True)
            continue
        r = json.loads(line[-1])
        print(f"{name:28s} {r['worst']:10.3e} {r['trained']:10.3e} {r['xyz15']:10.3e} {r['fine_ms']:8.3f}  {r['where']}", flush=True)


if __name__ == "__main__":
    main()
