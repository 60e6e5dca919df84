#!/bin/bash
# Builds the gfx950 engine library in-tree: trg-planner_amd/csrc/libtrg_engine.so
# (hipcc cross-compiles without a GPU).  -ffp-contract=off is part of the correctness contract:
# every fp32 decision must round exactly like the reference's non-FMA x86-64 build.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-function"
mkdir -p "$HERE/_obj"
# (the kernels at -O2: measured 0.15-0.2 ms per C3 build faster than -O3, whose extra unrolling only adds register pressure)
"$HIPCC" $FLAGS -O2 -c "$HERE/trg_kernels.hip" -o "$HERE/_obj/trg_kernels.o" "$@"
"$HIPCC" $FLAGS -c "$HERE/trg_engine.cpp" -o "$HERE/_obj/trg_engine.o"
# the voxel filter pulls in rocPRIM's radix sort (slow to compile): rebuilt only when it changed
if [ ! -f "$HERE/_obj/trg_voxel.o" ] || [ "$HERE/trg_voxel.hip" -nt "$HERE/_obj/trg_voxel.o" ] || [ "$HERE/trg_kernels.h" -nt "$HERE/_obj/trg_voxel.o" ]; then
  "$HIPCC" $FLAGS -c "$HERE/trg_voxel.hip" -o "$HERE/_obj/trg_voxel.o"
fi
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$HERE/_obj/trg_kernels.o" "$HERE/_obj/trg_engine.o" "$HERE/_obj/trg_voxel.o" -o "$HERE/libtrg_engine.so"
echo "built $HERE/libtrg_engine.so"
# pybind11 module trg_planner._trg_pybind: the reference's Python surface (TRG, Edge, NodeState, Node)
# over include/trg_shim.hpp, i.e. over the C ABI of the library above
PY="${PYTHON:-python3}"
PYINC="$("$PY" -c "import sysconfig; print(sysconfig.get_paths()['include'])")"
PBINC="$("$PY" -c "import pybind11; print(pybind11.get_include())")"
EXT="$("$PY" -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")"
g++ -O2 -std=c++17 -fPIC -shared -fvisibility=hidden -I"$PYINC" -I"$PBINC" "$HERE/trg_pybind.cpp" \
  -L"$HERE" -ltrg_engine -Wl,-rpath,'$ORIGIN/../csrc' -o "$HERE/../trg_planner/_trg_pybind$EXT"
echo "built $HERE/../trg_planner/_trg_pybind$EXT"
