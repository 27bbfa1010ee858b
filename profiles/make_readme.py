#!/usr/bin/env python3
"""Regenerate the current round's section of profiles/README.md from the reduced files of `make_summary.py`:
   python profiles/make_readme.py r02
Everything in the section is read from profiles/<tag>_*.json; the older rounds' sections below it are kept."""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
here = os.path.dirname(os.path.abspath(__file__))
J = lambda f: json.loads(open(os.path.join(here, f"{tag}_{f}.json")).read().strip().splitlines()[-1])
s = json.load(open(os.path.join(here, f"{tag}_summary.json")))
cl, st, tr, sq = s["launch_clusters"], s["step_traffic"], s["traffic"], s["sq"]
g = lambda k: round(cl[k]["avg_us"] / 1e3, 3)
bt, br, c1, c4 = J("bench_train"), J("bench_render"), J("bench_cfg3_r1024"), J("bench_cfg3_r4096")
c1g = J("bench_cfg3_r1024_graph")
bc = json.load(open(os.path.join(here, f"{tag}_bench_configs.json")))
K0, K1 = "nfl_render_kernel<3, 1, 10, 0>", "nfl_render_kernel<3, 1, 10, 1>"
F0, F1 = K0 + "|fine_pass_128_samples", K1 + "|fine_pass_128_samples"
frac = lambda k: 0.622e12 / cl[k]["avg_us"] / 1e6 / 2500
txt = f'''# profiles/ — MI355X, 1 GPU, ROCm 7.2, rocprofv3

All numbers: configs[1] = 4096 rays x (64 + 64) samples, base NeRF coarse + fine,
perturb 1, noise 1, white background, precision `f16x3` forward (the mode that meets the
1e-4 parity bar) + loss-scaled `f16` backward. `bash profiles/collect.sh <tag>` runs the four rocprofv3 passes
(`--kernel-trace --stats` over the full default `python3 bench.py --no-cpu-baseline --steps 20 --warmup 5`, then
`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc <SQ block>`, each its own run, over
`python3 bench.py --no-extras --steps 20 --warmup 5 --render-steps 3`: only train steps plus three forward-only
steps for the inference instantiation); `python profiles/make_summary.py gpurun_out/prof_<tag> <tag>` reduces them
to the files here and `python profiles/make_readme.py <tag>` writes this section from them. Boxes of the pool differ
by up to +-5 % on the MFMA-bound forward (MI355X_MICROARCH.md: devices hold different clocks under load), +-1 % on the
HBM-bound backward.

## {tag} — end of round 2

Same kernels as r01g plus: fp16 range tracking in the activation epilogue, camera prologue, NerfWLoss epilogue,
status word, ATen-order sampler, latent-table scatter in dgrad; cold-path kernel arguments re-read from the kernarg
segment (DESIGN.md section 3). Same-box A/B against the round-1 tree (both libraries built side by side,
`tests/time_passes.py` and `bench.py` alternated on one box): forward 1.51 vs 1.50 ms (inference), 1.73 vs 1.72
(training), dgrad 0.71 vs 0.75, wgrad 1.04 vs 1.05, whole step 5.20 vs 5.25 ms.
Then the backward lost the stash of `xyz_encoding_final`'s output and of the gradient w.r.t. it (the layer is linear:
its weight gradients and those of the 256 columns that read it are composed from `G = sum delta_dirh (x) h8`, DESIGN.md
section 5) and the sigma head's job merged into the G job's `h8` stream. Same-box A/B against the layout before
(`git worktree` of the previous commit built side by side): training forward 1.717 -> 1.695 ms, dgrad 0.695 -> 0.669,
wgrad 1.050 -> 0.944, train step **5.21 -> 4.98 ms**; configs[3] shape 6.02 -> 5.91 ms (4096 rays), 1.94 -> 1.85 ms (1024);
HBM bytes per step 18.3 -> 16.4 GB.
Last, the layer left the forward and dgrad streams altogether (folded into the two layers that read it at pack time,
DESIGN.md section 3): forward 1.52 -> 1.37 ms per fine pass, training forward 1.69 -> 1.52, dgrad 0.67 -> 0.63, train step
4.95 -> 4.66 ms, with the same worst parity error (1.29e-5).

| file | what |
|---|---|
| `{tag}_bench_train.json` | `python bench.py` (default = train step): **{bt["value"]:.3e} ray-samples/s**, {bt["ms_per_step"]:.2f} ms/step on this box (4.6-4.8 across the boxes seen since the layer was folded); forward-only {bt["render_only_value"]:.2e}; CPU oracle train step {bt["cpu_baseline"]["value"]:.2e} (16 threads; forward alone {bt["cpu_baseline"]["forward_value"]:.2e}) |
| `{tag}_bench_render.json` | `python bench.py --mode render`: **{br["value"]:.3e} ray-samples/s**, {br["ms_per_step"]:.2f} ms/step |
| `{tag}_bench_cfg3_r1024.json`, `{tag}_bench_cfg3_r1024_graph.json`, `{tag}_bench_cfg3_r4096.json` | `--workload cfg3` (configs[3] per-GPU shape: NeRF-W a+t, N_vocab 1500, per-ray near/far): at the README batch of 1024 rays **{c1["ms_per_step"]:.2f} ms/step = {c1["value"]:.2e}** eager, {c1g["ms_per_step"]:.2f} ms replayed from one HIP graph (`--graph`); the eager step had stayed at 1.87-1.95 ms when the device side came down to ~1.6 ms, until ten module-tree walks per step (0.7 ms of host time) were removed; {c4["ms_per_step"]:.2f} ms = {c4["value"]:.2e} at 4096 rays, where eager and graph are equal |
| `{tag}_bench_configs.json` | `tests/bench_configs.py`: configs[2] full NeRF-W train step {bc["cfg3_nerfw_train"]["ms_per_step"]:.2f} ms = {bc["cfg3_nerfw_train"]["ray_samples_per_s"]:.2e}; configs[4]-like eval (NeRF-W, 128+128, `test_time`, 131072 rays) {bc["cfg5_eval_direct"]["ms_per_131072_rays"]:.0f} ms = {bc["cfg5_eval_direct"]["rays_per_s"]:.2e} rays/s direct, {bc["cfg5_eval_hip_graph"]["ms_per_131072_rays"]:.0f} ms HIP-graph replayed; one 800 x 800 frame from (pose, intrinsics) {bc["cfg5_frame_800x800_camera_prologue"]["ms_per_frame"]:.0f} ms with the rays generated in the kernel prologue, {bc["cfg5_frame_800x800_ray_matrix"]["ms_per_frame"]:.0f} ms from a materialised ray matrix (the 21 MB of rays were never the cost) |
| `{tag}_train_step_kernel_stats.csv` | per-kernel time of the profiled command |
| `{tag}_train_step_pmc_{{FETCH_SIZE,WRITE_SIZE,SQ}}.csv` | PMC passes (nfl_* kernels) |
| `{tag}_train_step_kernel_launch_clusters.csv` | kernel trace split into the coarse-pass (64 samples/ray) and fine-pass (128) launch populations; the fine-pass `{K1}` launches average {g(F1)} ms = the launch `bench.py`'s `roofline.launch_ms` times with HIP events ({bt["roofline"]["launch_ms"]:.3f} ms); `{K0}`: {g(F0)} vs {bt["roofline_inference"]["launch_ms"]:.3f} ms |
| `{tag}_summary.json` | the reduction (`traffic`: HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md; `sq.mfma_busy_frac`; `step_traffic`: bytes of one whole train step) — `bench.py` reads `roofline.traffic` and `step_traffic` from it |
| `{tag}_time_passes.txt` | `tests/time_passes.py`: each kernel of the fine pass timed alone |
| `{tag}_mfma_shape_probe2.txt` | `profiles/tools/mfma_shape_probe2.hip`: the `32x32x16` and `16x16x32` fp16 MFMA shapes under the issue load of the render kernel (LDS reads, LDS-DMA pieces, epilogue VALU, barrier): -15.5 % time bare, -10.5 % render-like — the basis of DESIGN.md section 9 |
| `{tag}_mfma16_ablation.txt` | the real render kernel with every `32x32x16` MFMA issued as two `16x16x32` (`make variant VFLAGS=-DNFL_ABL_MFMA16`, timing only), alternated with the mainline on one box: inference forward 1.587 / 1.566 vs 1.529 / 1.523 ms (+3 %), training forward +0.7 % — the shape change does not pay in this kernel |
| `tools/fp16_probe.hip`, `tools/trapsts_probe.hip` | what gfx950 does beyond fp16's range (cvt -> inf, `x - inf` -> -inf, MFMA `inf * 0` / `inf - inf` -> 0xFFC00000, relu on the bit pattern -> 0) and that the sticky exception bits stay clear: why the range check is explicit (DESIGN.md section 3) |

Per launch, fine pass (4096 rays x 128 samples), this box:

| kernel | time | bound | achieved | MFMA pipe busy (PMC) | HBM bytes (PMC) |
|---|---|---|---|---|---|
| `{K0}` (inference forward) | {g(F0)} ms | mfma | {0.622e12/cl[F0]["avg_us"]/1e6:.0f} TFLOP/s algorithmic = frac {frac(F0):.3f} of 2.5 PFLOP/s (3 fp16 products issued per algorithmic product) | {100*sq[K0]["mfma_busy_frac"]:.1f} % | {tr[K0]["hbm_bytes_max_launch"]/1e6:.0f} MB (algorithmic ~7 MB) |
| `{K1}` (training forward: + fp16 stash + relu-mask words + loss epilogue) | {g(F1)} ms | mfma | frac {frac(F1):.3f} | {100*sq[K1]["mfma_busy_frac"]:.1f} % | {tr[K1]["hbm_bytes_max_launch"]/1e9:.2f} GB (stash write 2.7 GB) |
| `nfl_dgrad_kernel<10>` | {g("nfl_dgrad_kernel<10>|fine_pass_128_samples")} ms | hbm (stash write) | {tr["nfl_dgrad_kernel<10>"]["hbm_bytes_max_launch"]/cl["nfl_dgrad_kernel<10>|fine_pass_128_samples"]["avg_us"]/1e6:.2f} TB/s | {100*sq["nfl_dgrad_kernel<10>"]["mfma_busy_frac"]:.1f} % | {tr["nfl_dgrad_kernel<10>"]["hbm_bytes_max_launch"]/1e9:.2f} GB |
| `nfl_wgrad_kernel` | {g("nfl_wgrad_kernel|fine_pass_128_samples")} ms | hbm | {tr["nfl_wgrad_kernel"]["hbm_bytes_max_launch"]/cl["nfl_wgrad_kernel|fine_pass_128_samples"]["avg_us"]/1e6:.2f} TB/s of 8 TB/s | {100*sq["nfl_wgrad_kernel"]["mfma_busy_frac"]:.1f} % | {tr["nfl_wgrad_kernel"]["hbm_bytes_max_launch"]/1e9:.2f} GB |
| `nfl_compbwd_kernel` / `nfl_sample_pdf_kernel` / `nfl_pack_kernel` / `nfl_adam_kernel` | 14 / 16 / 8 / 17 us | hbm | — | — | 46 / 5 / 8 / 33 MB |

One train step: **{st["hbm_bytes"]/1e9:.1f} GB of HBM traffic (PMC) against {st["algorithmic_bytes"]/1e6:.0f} MB algorithmic** ({", ".join(f"{k} {v/1e9:.2f} GB" for k, v in st["by_kernel_bytes"].items() if v > 1e8)}): the fp16 activation and gradient stashes of the layer-major backward; DESIGN.md section 5
explains why they stay.

'''
path = os.path.join(here, "README.md")
old = open(path).read()
open(path, "w").write(txt + old[old.index("## r01g"):])
print("wrote", path)
