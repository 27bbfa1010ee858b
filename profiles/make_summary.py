#!/usr/bin/env python3
"""Reduce the raw rocprofv3 output of profiles/collect.sh to the files committed under profiles/:
   python profiles/make_summary.py gpurun_out/prof_<tag> <tag>
writes <tag>_train_step_kernel_stats.csv (copy), <tag>_train_step_pmc_{FETCH_SIZE,WRITE_SIZE,SQ}.csv
(nfl_* kernels only), <tag>_train_step_kernel_launch_clusters.csv and <tag>_summary.json.
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE is doubled on gfx950 as
MI355X_MICROARCH.md prescribes.  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs
x 256 CUs x 4 SIMDs)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# third argument: where to write (default: profiles/).  profiles/collect.sh reduces on the GPU box into gpurun_out/<tag>_reduced,
# because the raw rocprofv3 output (~80 MB) is more than gpurun copies back; the reduced files are then copied here.
here = sys.argv[3] if len(sys.argv) > 3 else os.path.dirname(os.path.abspath(__file__))
os.makedirs(here, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return hits[0]


def short(name):
    return name.replace("void ", "").split("(")[0]


# ---- kernel stats
stats_csv = one("stats/**/*kernel_stats.csv")
shutil.copy(stats_csv, os.path.join(here, f"{tag}_train_step_kernel_stats.csv"))
kernel_stats = {}
for r in csv.DictReader(open(stats_csv)):
    kernel_stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                      "pct": float(r["Percentage"])}
kernel_stats = dict(list(kernel_stats.items())[:12])

# ---- launch clusters from the trace (coarse pass = 64 samples / ray, fine = 128: two populations per kernel)
trace_csv = one("stats/**/*kernel_trace.csv")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace_csv)):
    n = short(r["Kernel_Name"])
    if n.startswith("nfl_"):
        dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(os.path.join(here, f"{tag}_train_step_kernel_launch_clusters.csv"), "w") as f:
    f.write("kernel,cluster,launches,avg_us,min_us,max_us\n")
    clusters = {}
    for n, d in dur.items():
        d = sorted(d)
        split = None
        if d[-1] > 1.5 * d[0] and len(d) >= 8:          # two populations: cut at the largest gap
            gaps = [(d[i + 1] - d[i], i) for i in range(len(d) - 1)]
            split = max(gaps)[1] + 1
        groups = [("all", d)] if split is None else [("coarse_pass_64_samples", d[:split]), ("fine_pass_128_samples", d[split:])]
        for g, v in groups:
            f.write(f"{n.replace(',', ';')},{g},{len(v)},{sum(v) / len(v):.1f},{v[0]:.1f},{v[-1]:.1f}\n")
            clusters[f"{n}|{g}"] = {"launches": len(v), "avg_us": sum(v) / len(v)}

# ---- PMC passes
def pmc(sub, outname):
    path = one(f"{sub}/**/*counter_collection.csv")
    rows = [r for r in csv.DictReader(open(path)) if "nfl_" in r["Kernel_Name"]]
    with open(os.path.join(here, outname), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                          "VGPR_Count", "Accum_VGPR_Count", "Counter_Name", "Counter_Value"],
                           extrasaction="ignore")
        w.writeheader()
        for r in rows:
            r = dict(r)
            r["Kernel_Name"] = short(r["Kernel_Name"])
            w.writerow(r)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


fetch = pmc("fetch", f"{tag}_train_step_pmc_FETCH_SIZE.csv")
write = pmc("write", f"{tag}_train_step_pmc_WRITE_SIZE.csv")
sq = pmc("sq", f"{tag}_train_step_pmc_SQ.csv")
traffic = {}
for k in fetch:
    fk, wk = max(fetch[k]["FETCH_SIZE"]), max(write[k]["WRITE_SIZE"]) if k in write else 0.0
    traffic[k] = {"FETCH_SIZE_KB_max_launch": fk, "WRITE_SIZE_KB_max_launch": wk,
                  "hbm_bytes_max_launch": (2.0 * fk + wk) * 1024.0}
sqs = {}
for k, v in sq.items():
    m = {c: max(x) for c, x in v.items()}          # the fine-pass launch (largest) of each kernel
    if m.get("GRBM_GUI_ACTIVE"):
        m["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    sqs[k] = m
# ---- HBM bytes of one whole train step: all nfl_* dispatches of the PMC passes (warm-up + timed steps of
# `bench.py --no-extras`, identical work each) divided by the number of steps (= nfl_adam_kernel dispatches)
# -- counted per PASS: FETCH_SIZE and WRITE_SIZE come from two separate runs, and a run's time-based sustained loop does not
# execute the same number of steps twice
n_steps = max(1, len(fetch.get("nfl_adam_kernel", {}).get("FETCH_SIZE", [])))
n_steps_w = max(1, len(write.get("nfl_adam_kernel", {}).get("WRITE_SIZE", [])))
# the inference instantiation `<..., 0>` belongs to the forward-only companion steps (--render-steps), not to a train step;
# their few nfl_pack / nfl_sample_pdf launches (28 MB each) are left in: 3 MB per step of 18 GB
in_step = lambda k: not (k.startswith("nfl_render_kernel") and k.rstrip().endswith(", 0>"))
tot_f = sum(sum(v["FETCH_SIZE"]) for k, v in fetch.items() if in_step(k))
tot_w = sum(sum(v["WRITE_SIZE"]) for k, v in write.items() if in_step(k))
R, S, F, NPARAM = 4096, 64, 128, 2 * 595844
algorithmic = {
    "rays_ts_in": R * (32 + 8),
    "outputs": R * (4 * (S + F) + 2 * (4 + 12 + 4)),                 # weights_*, opacity, rgb, depth (808 B/ray)
    "grad_outputs_in": R * (4 * (S + F) + 2 * (4 + 12 + 4)),
    "parameters_read_fwd_bwd": 2 * NPARAM * 4,
    "gradients_written": NPARAM * 4,
    "adam_read_write": NPARAM * 28,                                    # p, g, m, v in; p, m, v out
}
step_traffic = {"hbm_bytes": (2.0 * tot_f / n_steps + tot_w / n_steps_w) * 1024.0, "steps_counted": n_steps, "steps_counted_write_pass": n_steps_w,
                "algorithmic_bytes": float(sum(algorithmic.values())), "algorithmic_breakdown": algorithmic,
                "by_kernel_bytes": {k: (2.0 * sum(fetch[k]["FETCH_SIZE"]) / n_steps + sum(write.get(k, {}).get("WRITE_SIZE", [0.0])) / n_steps_w)
                                    * 1024.0 for k in fetch if in_step(k)},
                "note": "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB summed over every nfl_* dispatch of one train step; the "
                        "excess over the algorithmic bytes is the fp16 activation / gradient stashes of the layer-major "
                        "backward (DESIGN.md section 5)"}
import datetime
import hashlib
lib = os.path.join(root, "nerf_fl_amd", "libnerf_fl_amd.so")
provenance = {"tag": tag, "collected_utc": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
              "lib_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None,
              "command": "profiles/collect.sh " + tag}
json.dump({"provenance": provenance, "kernel_stats": kernel_stats, "launch_clusters": clusters, "traffic": traffic, "sq": sqs, "step_traffic": step_traffic},
          open(os.path.join(here, f"{tag}_summary.json"), "w"), indent=1)
for f in ("bench_train.json", "bench_render.json", "time_passes.txt", "bench_configs.json", "bench_cfg3_r1024.json",
          "bench_cfg3_r4096.json", "bench_cfg3_r1024_graph.json"):
    p = os.path.join(src, f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(here, f"{tag}_{f}"))
print("wrote", tag, "summary;", len(traffic), "kernels with traffic,", len(sqs), "with SQ counters")
