"""trg_planner -- MI355X-native drop-in for the hot path of ziwon-park/TRG-planner.

Mirrors the reference's pybind11 module of the same name
(/root/reference/python/trg_planner/pybind/trg_planner_pybind.cpp:19-78): ``TRG``, ``Edge``,
``NodeState``, ``Node`` and ``TRGPlanner``, over the C ABI in include/trg_engine.h.
"""
__version__ = "1.0.0"

from ._engine import (Engine, CsrGraph, TrgError, build_library, load_library, LIB_PATH)  # noqa: F401
from .api import TRG, Edge, Node, NodeState  # noqa: F401
from .planner import TRGPlanner  # noqa: F401
