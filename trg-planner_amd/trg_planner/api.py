"""The reference's ``TRG`` / ``Node`` / ``Edge`` / ``NodeState`` Python surface.

Reference: python/trg_planner/pybind/trg_planner_pybind.cpp:23-43 and
cpp/trg_planner/core/trg_planner/include/graph/trg.h:18-98.  The classes are the compiled pybind11
module ``_trg_pybind`` (csrc/trg_pybind.cpp) over the C++ shim ``include/trg_shim.hpp``, i.e. over
the C ABI of ``libtrg_engine.so``; nothing is re-implemented in Python.  There is no fallback: a
missing extension raises ImportError.
"""
from __future__ import annotations

try:
    from ._trg_pybind import TRG, Edge, Node, NodeState  # noqa: F401
except ImportError as exc:  # pragma: no cover
    raise ImportError("trg_planner._trg_pybind is missing: run trg-planner_amd/csrc/build.sh "
                      "(__graft_entry__.build()); there is no Python fallback") from exc
