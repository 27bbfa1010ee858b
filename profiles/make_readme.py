#!/usr/bin/env python3
"""Regenerate the current round's section of profiles/README.md from the reduced files of `make_summary.py`:
   python profiles/make_readme.py r02
Everything in the section is read from profiles/<tag>_*.json; the older rounds' sections below it are kept."""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
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
(`--kernel-trace --stats` over `python3 bench.py --no-cpu-baseline --only-default-backward --sustained-seconds 0 --steps 20 --warmup 5`, then
`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc <SQ block>`, each its own run, over
`python3 bench.py --no-extras --steps 20 --warmup 5 --render-steps 3`: only train steps plus three forward-only
steps for the inference instantiation) and reduces them on the box (`profiles/make_summary.py`: the raw output is more than
gpurun copies back) into `gpurun_out/<tag>_reduced/`, whose files are copied here; `python profiles/make_readme.py <tag>`
writes this section from them. Boxes of the pool differ
by up to +-5 % on the MFMA-bound forward (MI355X_MICROARCH.md: devices hold different clocks under load), +-1 % on the
HBM-bound backward.

## {tag} — end of round 3

What changed against r02 in the kernels: the backward has three arithmetics (DESIGN.md section 5) — the default `f16` kernels
are round 2's (`nfl_dgrad_kernel<10, 1, 1>`, `nfl_wgrad_kernel<false>`; wgrad now addresses its stream with scalar bases),
`f16w` adds the dgrad instantiation on hi + lo weight fragments (`<10, 1, 2>`), `f16x3` the split-operand dgrad (`<10, 2, 2>`), the
training forward that also stores its residual operands (`nfl_render_kernel<3, 1, 10, 3>`) and a one-pass split wgrad
(`nfl_wgrad_kernel<true>`); encoder widths 1..15 / 1..4 run in the two existing instantiations; the camera of a ray-generating
pass can live in device memory; the weight gradients are written in place into one flat gradient arena. The profiled command is the
default (`f16` backward); `{tag}_time_passes.txt` times the kernels of all three arithmetics alone, `{tag}_bench_train.json`
carries all three step figures. Evidence of this round that is not a profile: `r03_psnr_backward_attribution.txt` (which product
of the backward offsets the NeRF-W training curve: the fp16-rounded transposed weights of the gradient chain) and
`r03_prods_sweep.txt` (no layer of the forward can give up one of its three products).

| file | what |
|---|---|
| `{tag}_bench_train.json` | `python bench.py` (default = train step, `f16` backward): **{bt["value"]:.3e} ray-samples/s**, {bt["ms_per_step"]:.2f} ms/step on this box (sustained over {bt["sustained"]["steps"]} event-timed steps: median {bt["sustained"]["median_ms"]:.2f}, p90 {bt["sustained"]["p90_ms"]:.2f} ms); `f16w` (gradient chain on hi + lo weights) {bt["exact_weight_chain"]["ms_per_step"]:.2f} ms = {bt["value_exact_weight_chain"]:.3e}; `f16x3` (fp32-class) {bt["fp32_class"]["ms_per_step"]:.2f} ms = {bt["value_fp32_class"]:.3e}; forward-only {bt["render_only_value"]:.2e}; CPU oracle train step {bt["cpu_baseline"]["value"]:.2e} (16 threads; forward alone {bt["cpu_baseline"]["forward_value"]:.2e}) |
| `{tag}_bench_render.json` | `python bench.py --mode render`: **{br["value"]:.3e} ray-samples/s**, {br["ms_per_step"]:.2f} ms/step |
| `{tag}_bench_cfg3_r1024.json`, `{tag}_bench_cfg3_r1024_graph.json`, `{tag}_bench_cfg3_r4096.json` | `--workload cfg3` (configs[3] per-GPU shape: NeRF-W a+t, N_vocab 1500, per-ray near/far): at the README batch of 1024 rays **{c1["ms_per_step"]:.2f} ms/step = {c1["value"]:.2e}** eager, {c1g["ms_per_step"]:.2f} ms replayed from one HIP graph (`--graph`); {c4["ms_per_step"]:.2f} ms = {c4["value"]:.2e} at 4096 rays |
| `{tag}_bench_configs.json` | `tests/bench_configs.py`: configs[2] full NeRF-W train step {bc["cfg3_nerfw_train"]["ms_per_step"]:.2f} ms = {bc["cfg3_nerfw_train"]["ray_samples_per_s"]:.2e}; configs[4]-like eval (NeRF-W, 128+128, `test_time`, 131072 rays) {bc["cfg5_eval_direct"]["ms_per_131072_rays"]:.0f} ms = {bc["cfg5_eval_direct"]["rays_per_s"]:.2e} rays/s direct, {bc["cfg5_eval_hip_graph"]["ms_per_131072_rays"]:.0f} ms HIP-graph replayed; one 800 x 800 frame from (pose, intrinsics) {bc["cfg5_frame_800x800_camera_prologue"]["ms_per_frame"]:.0f} ms with the rays generated in the kernel prologue, {bc["cfg5_frame_800x800_ray_matrix"]["ms_per_frame"]:.0f} ms from a materialised ray matrix, {bc["cfg5_frame_800x800_camera_prologue_hip_graph"]["ms_per_frame"]:.0f} ms with the chunk replayed from ONE HIP graph whose prologue reads the camera from device memory ({bc["cfg5_frame_800x800_camera_prologue_hip_graph"]["captures"]} capture) |
| `{tag}_train_step_kernel_stats.csv` | per-kernel time of the profiled command |
| `{tag}_train_step_pmc_{{FETCH_SIZE,WRITE_SIZE,SQ}}.csv` | PMC passes (nfl_* kernels) |
| `{tag}_train_step_kernel_launch_clusters.csv` | kernel trace split into the coarse-pass (64 samples/ray) and fine-pass (128) launch populations; the fine-pass `{K1}` launches average {g(F1)} ms = the launch `bench.py`'s `roofline.launch_ms` times with HIP events ({bt["roofline"]["launch_ms"]:.3f} ms); `{K0}`: {g(F0)} vs {bt["roofline_inference"]["launch_ms"]:.3f} ms |
| `{tag}_summary.json` | the reduction (`traffic`: HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md; `sq.mfma_busy_frac`; `step_traffic`: bytes of one whole train step) — `bench.py` reads `roofline.traffic` and `step_traffic` from it |
| `{tag}_time_passes.txt` | `tests/time_passes.py`: each kernel of the fine pass timed alone |
| `r03_psnr_backward_attribution.txt` | 6-8 HIP fits per row of the NeRF-W parity scene against the stored reference runs, with the default library and with variant builds that change ONE ingredient of the backward (csrc/nfl_diag.h): the training-curve offset follows the fp16-rounded transposed weights of the gradient chain and nothing else |
| `r03_prods_sweep.txt` | `tests/report_prods.py`: the x3 render kernel rebuilt 24 times with one layer group's one correction product dropped, worst output error over all two-pass fixtures + launch time: no trunk layer / density head / transient branch can give up a product |
| `tools/kres.py` | VGPR / AGPR / SGPR / spill / scratch of every kernel in the built library, from the code object's notes (no GPU) |
| `tools/fp16_probe.hip`, `tools/trapsts_probe.hip` | what gfx950 does beyond fp16's range (cvt -> inf, `x - inf` -> -inf, MFMA `inf * 0` / `inf - inf` -> 0xFFC00000, relu on the bit pattern -> 0) and that the sticky exception bits stay clear: why the range check is explicit (DESIGN.md section 3) |

Per launch, fine pass (4096 rays x 128 samples), this box:

| kernel | time | bound | achieved | MFMA pipe busy (PMC) | HBM bytes (PMC) |
|---|---|---|---|---|---|
| `{K0}` (inference forward) | {g(F0)} ms | mfma | {0.622e12/cl[F0]["avg_us"]/1e6:.0f} TFLOP/s algorithmic = frac {frac(F0):.3f} of 2.5 PFLOP/s (3 fp16 products issued per algorithmic product) | {100*sq[K0]["mfma_busy_frac"]:.1f} % | {tr[K0]["hbm_bytes_max_launch"]/1e6:.0f} MB (algorithmic ~7 MB) |
| `{K1}` (training forward: + fp16 stash + relu-mask words + loss epilogue) | {g(F1)} ms | mfma | frac {frac(F1):.3f} | {100*sq[K1]["mfma_busy_frac"]:.1f} % | {tr[K1]["hbm_bytes_max_launch"]/1e9:.2f} GB (stash write 2.7 GB) |
| `nfl_dgrad_kernel<10, 1, 1>` | {g("nfl_dgrad_kernel<10, 1, 1>|fine_pass_128_samples")} ms | hbm (stash write) | {tr["nfl_dgrad_kernel<10, 1, 1>"]["hbm_bytes_max_launch"]/cl["nfl_dgrad_kernel<10, 1, 1>|fine_pass_128_samples"]["avg_us"]/1e6:.2f} TB/s | {100*sq["nfl_dgrad_kernel<10, 1, 1>"]["mfma_busy_frac"]:.1f} % | {tr["nfl_dgrad_kernel<10, 1, 1>"]["hbm_bytes_max_launch"]/1e9:.2f} GB |
| `nfl_wgrad_kernel<false>` | {g("nfl_wgrad_kernel<false>|fine_pass_128_samples")} ms | hbm | {tr["nfl_wgrad_kernel<false>"]["hbm_bytes_max_launch"]/cl["nfl_wgrad_kernel<false>|fine_pass_128_samples"]["avg_us"]/1e6:.2f} TB/s of 8 TB/s | {100*sq["nfl_wgrad_kernel<false>"]["mfma_busy_frac"]:.1f} % | {tr["nfl_wgrad_kernel<false>"]["hbm_bytes_max_launch"]/1e9:.2f} GB |
| `nfl_compbwd_kernel` / `nfl_sample_pdf_kernel` / `nfl_pack_kernel` / `nfl_adam_kernel` | 14 / 16 / 8 / 17 us | hbm | — | — | 46 / 5 / 8 / 33 MB |

One train step: **{st["hbm_bytes"]/1e9:.1f} GB of HBM traffic (PMC) against {st["algorithmic_bytes"]/1e6:.0f} MB algorithmic** ({", ".join(f"{k} {v/1e9:.2f} GB" for k, v in st["by_kernel_bytes"].items() if v > 1e8)}): the fp16 activation and gradient stashes of the layer-major backward; DESIGN.md section 5
explains why they stay.

'''
path = os.path.join(here, "README.md")
old = open(path).read()
keep = "## r02" if "## r02" in old and tag != "r02" else "## r01g"
open(path, "w").write(txt + old[old.index(keep):])
print("wrote", path)
