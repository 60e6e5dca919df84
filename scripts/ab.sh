#!/bin/bash
# Run on the GPU box: alternate the builds of the engine library found as ab/lib*.so on the same GPU
# and print ms_per_step / loop / deferred / finalize / set-map / index-kernel times of each run (same-box
# A/B measurements: boxes differ by a few percent, runs on one box by well under one).
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cat > /tmp/ab_fmt.py <<'PY'
import json, sys
d = json.load(sys.stdin)
b = d["breakdown_last_step"]
print(sys.argv[1], round(d["ms_per_step"], 2), round(b["ms_bfs_loop"], 2), round(b["ms_deferred"], 2),
      round(b["ms_finalize_host"], 2), round(b["ms_set_map_total"], 2), round(b["ms_index_build_gpu"], 2))
PY
for i in 1 2 3; do
  for f in "$R"/ab/lib*.so; do
    v=$(basename "$f" .so)
    TRG_ENGINE_LIB="$f" python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 /tmp/ab_fmt.py "$v"
  done
done
