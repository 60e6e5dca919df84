#!/bin/bash
# Run on the GPU box: stage-by-stage cost of the edge kernel (profiling builds with
# -DTRG_EDGE_STAGE_CUT=n; the product library is rebuilt at the end).
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/edge_mb"
mkdir -p "$OUT"
for cut in 0 1 2 3 4 5; do
  bash "$R/trg-planner_amd/csrc/build.sh" -DTRG_EDGE_STAGE_CUT=$cut > "$OUT/build_$cut.log" 2>&1
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t$cut" -- python3 "$R/scripts/edge_microbench.py" > "$OUT/run_$cut.log" 2>&1)
  f=$(ls "$OUT"/t$cut/*/*kernel_stats.csv | head -1)
  echo "cut=$cut $(grep k_edges "$f" | cut -d, -f1-5 | sed 's/void trg::(anonymous namespace):://') $(tail -1 "$OUT/run_$cut.log")"
done
bash "$R/trg-planner_amd/csrc/build.sh" > "$OUT/build_final.log" 2>&1
