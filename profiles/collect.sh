#!/bin/bash
# Run on the GPU box from the repo root:  bash profiles/collect.sh <tag>   (e.g. r02a)
# Four rocprofv3 passes over bench.py's default train step (configs[1]): kernel trace + stats over the FULL default
# command (so that the roofline launches bench.py times with events are in the trace), then FETCH_SIZE, WRITE_SIZE and
# the SQ block in separate --pmc passes (MI355X_MICROARCH.md, HBM / rocprofv3 section) over the same steps WITHOUT the
# companions (--no-extras: only warm-up + timed train steps, so that bytes summed over all dispatches / steps = bytes per
# step).  Raw output goes to gpurun_out/prof_<tag>/; profiles/make_summary.py reduces it.
set -e -o pipefail
TAG=${1:-r01b}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o tr -- python3 $ROOT/bench.py --no-cpu-baseline --only-default-backward --sustained-seconds 0 --steps 20 --warmup 5 > "$OUT/stats.log" 2>&1
CMD="python3 $ROOT/bench.py --no-extras --steps 20 --warmup 5 --render-steps 3"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o pmc -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o pmc -- $CMD > "$OUT/write.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d "$OUT/sq" -o pmc -- $CMD > "$OUT/sq.log" 2>&1
cd "$ROOT"
python3 bench.py > "$OUT/bench_train.json" 2> "$OUT/bench_train.err"
python3 bench.py --mode render --no-cpu-baseline > "$OUT/bench_render.json" 2> "$OUT/bench_render.err"
(python3 tests/time_passes.py f16x3; python3 tests/time_passes.py f16x3 --backward f16w; python3 tests/time_passes.py f16x3 --backward f16x3) > "$OUT/time_passes.txt" 2>&1
python3 tests/bench_configs.py > "$OUT/bench_configs.json" 2> "$OUT/bench_configs.err"
python3 bench.py --no-extras --workload cfg3 --rays 4096 --steps 20 --warmup 5 > "$OUT/bench_cfg3_r4096.json" 2>/dev/null
python3 bench.py --no-extras --workload cfg3 --rays 1024 --steps 100 --warmup 10 --graph > "$OUT/bench_cfg3_r1024_graph.json" 2>/dev/null
python3 bench.py --no-extras --workload cfg3 --rays 1024 --steps 100 --warmup 10 > "$OUT/bench_cfg3_r1024.json" 2>/dev/null
# reduce here (the raw output is more than gpurun copies back) and keep only the reduction
python3 profiles/make_summary.py "$OUT" "$TAG" "$ROOT/gpurun_out/${TAG}_reduced"
cp "$OUT"/*.log "$OUT"/*.err "$ROOT/gpurun_out/${TAG}_reduced/" 2>/dev/null || true
if [ -z "$NFL_KEEP_RAW" ]; then rm -rf "$OUT"; fi
ls "$ROOT/gpurun_out/${TAG}_reduced"
