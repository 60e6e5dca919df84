#!/bin/bash
# Run on the GPU box: alternate two builds of the engine library (ab/libA.so, ab/libB.so) on the same
# GPU and print ms_per_step / loop time of each run.
R="${GRAFT_REPO_ROOT:-$(pwd)}"
for i in 1 2 3; do
  for v in A B; do
    TRG_ENGINE_LIB="$R/ab/lib$v.so" python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.load(sys.stdin);b=d['breakdown_last_step'];print('$v', round(d['ms_per_step'],2), round(b['ms_bfs_loop'],2), round(b['ms_deferred'],2), round(b['ms_finalize_host'],2))"
  done
done
