#!/bin/bash
# Run on the GPU box: per-phase cycle counts of k_level_expand (profiling build -DLV_PHASE_TIMING;
# the product library is rebuilt at the end).
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
bash "$R/trg-planner_amd/csrc/build.sh" -DLV_PHASE_TIMING "$@" > /dev/null 2>&1
TRG_PHASE_TIMING=1 python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "expand phases" | tail -1
# the same for k_level_resolve (cycles per workgroup)
bash "$R/trg-planner_amd/csrc/build.sh" -DLV_PHASE_TIMING -DLV_PHASE_TIMING_RESOLVE "$@" > /dev/null 2>&1
TRG_PHASE_TIMING=r python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "resolve phases" | tail -1
bash "$R/trg-planner_amd/csrc/build.sh" > /dev/null 2>&1
