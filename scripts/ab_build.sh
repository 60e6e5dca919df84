#!/bin/bash
# Build a variant of the engine library for scripts/ab.sh: ab/lib<name>.so, kernels compiled with the
# given extra flags (e.g. -DLV_WAVES_PER_SIMD=6); host objects are taken from the current build.
# Usage: bash scripts/ab_build.sh <name> [extra hipcc flags...]
set -euo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
N="$1"; shift
C="$R/trg-planner_amd/csrc"
mkdir -p "$R/ab/_obj"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -O2 -c "$C/trg_kernels.hip" -o "$R/ab/_obj/k_$N.o" "$@"  # (-O2 as csrc/build.sh; a later -O3 in "$@" overrides)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$R/ab/_obj/k_$N.o" "$C/_obj/trg_engine.o" "$C/_obj/trg_voxel.o" -o "$R/ab/lib$N.so"
echo "built ab/lib$N.so"
