#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of one bench run, top kernels printed.
# Usage: bash scripts/kstats.sh <tag> [bench args...]
set -uo pipefail
TAG="${1:-x}"; shift || true
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/ks_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp TRG_BENCH_FAST=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "trace rc=$?"
cd "$R"
f=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
cp "$f" "$OUT/kernel_stats.csv"
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = r["Name"].replace("trg::(anonymous namespace)::", "").split("(")[0][:40]
    print(f"{n:40s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.3f} avg_us={float(r['AverageNs'])/1e3:9.2f} {r['Percentage']}%")
PY
