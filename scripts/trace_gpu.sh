#!/bin/bash
# kernel-trace --stats only (no PMC); usage: bash scripts/trace_gpu.sh <tag> [workload] [steps]
set -uo pipefail
TAG="${1:-t}"; WL="${2:-c3}"; STEPS="${3:-2}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/prof_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp TRG_BENCH_FAST=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --workload "$WL" --steps "$STEPS" --warmup 1 --no-cpu-baseline > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err"
echo "trace rc=$?"
cd "$R"; python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.md" 2>&1
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
cat "$OUT/summary.md" | head -45
