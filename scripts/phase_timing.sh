#!/bin/bash
# Run on the GPU box: per-phase cycle counts of the level kernels.  The profiling variants
# (-DLV_PHASE_TIMING / -DLV_PHASE_TIMING_RESOLVE) are built into ab/ and selected through
# TRG_ENGINE_LIB, like scripts/ab_build.sh does: the product library is never replaced.
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
bash "$R/scripts/ab_build.sh" phase_expand -DLV_PHASE_TIMING "$@" > /dev/null 2>&1
TRG_ENGINE_LIB="$R/ab/libphase_expand.so" TRG_PHASE_TIMING=1 python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "expand phases" | tail -1 || true
bash "$R/scripts/ab_build.sh" phase_resolve -DLV_PHASE_TIMING -DLV_PHASE_TIMING_RESOLVE "$@" > /dev/null 2>&1
TRG_ENGINE_LIB="$R/ab/libphase_resolve.so" TRG_PHASE_TIMING=r python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "resolve phases" | tail -1 || true
rm -f "$R/ab/libphase_expand.so" "$R/ab/libphase_resolve.so"
