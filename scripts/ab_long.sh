#!/bin/bash
# like ab.sh, 20 steps per run and 5 alternations (for differences below the run-to-run noise)
R="${GRAFT_REPO_ROOT:-$(pwd)}"
bash "$R/scripts/ab.sh" > /dev/null 2>&1
for i in 1 2 3 4 5; do
  for f in "$R"/ab/lib*.so; do
    v=$(basename "$f" .so)
    TRG_ENGINE_LIB="$f" python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 /tmp/ab_fmt.py "$v"
  done
done
