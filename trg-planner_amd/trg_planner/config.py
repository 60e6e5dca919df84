"""YAML parameters of the reference planner, same keys and per-key defaults as
``TRGPlanner::setParams`` (cpp/.../src/planner/trg_planner.cpp:103-129; keys documented in the
reference's config/README.md:52-71)."""
from __future__ import annotations

import yaml

DEFAULTS = {
    "isVerbose": True,
    "timer": {"graphRate": 1.0, "planningRate": 1.0},
    "map": {"isPrebuiltMap": False, "prebuiltMapPath": "", "isVoxelize": False, "voxelSize": 0.1},
    "trg": {"isPrebuiltTRG": False, "prebuiltTRGPath": "", "isUpdate": False, "expandDist": 0.6,
            "robotSize": 0.3, "sampleNum": 20, "heightThreshold": 0.15, "collisionThreshold": 0.2,
            "updateCollisionThreshold": 0.2, "safetyFactor": 1.0, "goalTolerance": 0.8},
}


class Params:
    """Flat attribute view with the reference's member names (planner/trg_planner.h param_)."""

    def __init__(self, cfg):
        g = lambda sec, key: (cfg.get(sec) or {}).get(key, DEFAULTS[sec][key])  # noqa: E731
        self.isVerbose = bool(cfg.get("isVerbose", DEFAULTS["isVerbose"]))
        self.graph_rate = float(g("timer", "graphRate"))
        self.planning_rate = float(g("timer", "planningRate"))
        self.isPreMap = bool(g("map", "isPrebuiltMap"))
        self.preMapPath = str(g("map", "prebuiltMapPath"))
        self.isVoxelize = bool(g("map", "isVoxelize"))
        self.VoxelSize = float(g("map", "voxelSize"))
        self.isPreGraph = bool(g("trg", "isPrebuiltTRG"))
        self.preGraphPath = str(g("trg", "prebuiltTRGPath"))
        self.isUpdate = bool(g("trg", "isUpdate"))
        self.expandDist = float(g("trg", "expandDist"))
        self.robotSize = float(g("trg", "robotSize"))
        self.sampleNum = int(g("trg", "sampleNum"))
        self.heightThreshold = float(g("trg", "heightThreshold"))
        self.collisionThreshold = float(g("trg", "collisionThreshold"))
        self.updateCollisionThreshold = float(g("trg", "updateCollisionThreshold"))
        self.safetyFactor = float(g("trg", "safetyFactor"))
        self.goal_tolerance = float(g("trg", "goalTolerance"))


def load_params(config_path) -> Params:
    with open(config_path) as f:
        cfg = yaml.safe_load(f) or {}
    return Params(cfg)
