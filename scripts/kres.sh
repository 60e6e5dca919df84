#!/bin/bash
# Register / LDS / spill numbers of the level kernels in the built object (no GPU needed).
# Usage: bash scripts/kres.sh [object]   (default: trg-planner_amd/csrc/_obj/trg_kernels.o)
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
O="${1:-$R/trg-planner_amd/csrc/_obj/trg_kernels.o}"
T=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
objcopy -O binary --only-section=.hip_fatbin "$O" "$T/fb.bin"
$B/clang-offload-bundler --unbundle --type=o --input="$T/fb.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$T/k.co"
$B/llvm-readelf --notes "$T/k.co" | awk '
  /\.group_segment_fixed_size:/ {lds=$2}
  /\.name:/ {name=$2}
  /\.sgpr_spill_count:/ {ss=$2}
  /\.vgpr_count:/ {v=$2}
  /\.vgpr_spill_count:/ {vs=$2; printf "%-70s lds=%6d vgpr=%3d sgpr_spill=%3d vgpr_spill=%3d\n", substr(name,1,70), lds, v, ss, vs}' \
  | sed 's/_ZN3trg12_GLOBAL__N_1[0-9]*//' | grep -E "${KRES_FILTER:-k_level|k_calls_gather}"
rm -rf "$T"
