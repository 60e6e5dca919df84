#!/bin/bash
# Run on the GPU box (inside gpurun): kernel-trace stats + separate PMC passes for HBM bytes.
# Usage: bash scripts/profile_gpu.sh <tag> [workload]
set -uo pipefail
TAG="${1:-r1}"
WL="${2:-c3}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp TRG_BENCH_FAST=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --updates 0 > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err"
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" --workload "$WL" --steps 1 --warmup 0 --no-cpu-baseline --updates 0 > "$OUT/bench_fetch.json" 2> "$OUT/bench_fetch.err"
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" --workload "$WL" --steps 1 --warmup 0 --no-cpu-baseline --updates 0 > "$OUT/bench_write.json" 2> "$OUT/bench_write.err"
echo "write rc=$?"
cd "$R"
python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.md" 2> "$OUT/summary.err"
# keep the merged-back payload small: drop the raw per-dispatch traces, keep stats + summary
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
find "$OUT" -name '*counter_collection.csv' -size +20M -delete
ls -la "$OUT" "$OUT"/*/* 2>/dev/null | head -40
