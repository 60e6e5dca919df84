#!/bin/bash
# Run on the GPU box: alternate variants of the engine on the same GPU and print ms_per_step / loop /
# deferred / finalize / set-map / index-kernel times of each run (same-box A/B measurements: boxes differ
# by a few percent, runs on one box by well under one).
#   bash scripts/ab.sh                      every ab/lib*.so, default environment
#   bash scripts/ab.sh name=lib[,VAR=val]…  named variants: library (file under ab/ or "product") + environment
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cat > /tmp/ab_fmt.py <<'PY'
import json, sys
d = json.load(sys.stdin)
b = d["breakdown_last_step"]
print(sys.argv[1], round(d["ms_per_step"], 2), round(b["ms_bfs_loop"], 2), round(b["ms_deferred"], 2),
      round(b["ms_finalize_host"], 2), round(b["ms_set_map_total"], 2), round(b["ms_index_build_gpu"], 2),
      b.get("presampled_nodes", ""), b.get("bfs_levels", ""))
PY
specs=("$@")
if [ ${#specs[@]} -eq 0 ]; then
  for f in "$R"/ab/lib*.so; do specs+=("$(basename "$f" .so)=$(basename "$f")"); done
fi
for i in 1 2 3; do
  for sp in "${specs[@]}"; do
    name="${sp%%=*}"; rest="${sp#*=}"
    IFS=',' read -ra parts <<< "$rest"
    lib="${parts[0]}"
    if [ "$lib" = "product" ]; then libpath="$R/trg-planner_amd/csrc/libtrg_engine.so"; else libpath="$R/ab/$lib"; fi
    envs=("TRG_ENGINE_LIB=$libpath" "TRG_BENCH_FAST=1")
    for ((k=1; k<${#parts[@]}; k++)); do envs+=("${parts[$k]}"); done
    env "${envs[@]}" timeout -k 10 "${AB_TIMEOUT:-150}" python3 "$R/bench.py" --steps "${AB_STEPS:-10}" --warmup 2 --no-cpu-baseline 2>/dev/null | python3 /tmp/ab_fmt.py "$name"
  done
done
