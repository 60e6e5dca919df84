#!/bin/bash
# Run on the GPU box: stage-by-stage cost of the edge kernel (profiling builds with
# -DTRG_EDGE_STAGE_CUT=n; the product library is rebuilt at the end).
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/edge_mb"
rm -rf "$OUT"
mkdir -p "$OUT"
for cut in 0 1 2 3 4 5; do
  bash "$R/trg-planner_amd/csrc/build.sh" -DTRG_EDGE_STAGE_CUT=$cut > "$OUT/build_$cut.log" 2>&1
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t$cut" -- python3 "$R/scripts/edge_microbench.py" > "$OUT/run_$cut.log" 2>&1)
  f=$(ls -t "$OUT"/t$cut/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$cut" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_edges" in r["Name"]:
        print(f"cut={sys.argv[2]} k_edges calls={r['Calls']} total_ms={int(r['TotalDurationNs']) / 1e6:.3f}")
PY
done
bash "$R/trg-planner_amd/csrc/build.sh" > "$OUT/build_final.log" 2>&1
