"""GPU: BASELINE config 1 plumbing -- indoor-style .pcd + YAML -> TRGPlanner (FSM threads, command
channel) -> 'graph expand' -> goal -> path, through the mirrored reference API."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_planner_plumbing_indoor(tmp_path, synth):
    import trg_planner
    from trg_planner import pcd
    pts, boxes = synth.indoor_cloud(seed=1, size=(16.0, 12.0), n_boxes=5)
    (tmp_path / "prebuilt_maps").mkdir()
    (tmp_path / "config").mkdir()
    pcd.write_pcd(tmp_path / "prebuilt_maps" / "sim_indoor_0.1.pcd", pts, "binary")
    (tmp_path / "config" / "indoor.yaml").write_text(
        "isVerbose: false\ntimer:\n  graphRate: 50.0\n  planningRate: 50.0\nmap:\n  isPrebuiltMap: true\n"
        "  prebuiltMapPath: \"prebuilt_maps/sim_indoor_0.1.pcd\"\n  isVoxelize: true\n  voxelSize: 0.2\n"
        "trg:\n  isPrebuiltTRG: false\n  isUpdate: false\n  expandDist: 0.4\n  robotSize: 0.3\n"
        "  sampleNum: 15\n  heightThreshold: 0.15\n  collisionThreshold: 0.1\n"
        "  updateCollisionThreshold: 0.1\n  safetyFactor: 3.0\n  goalTolerance: 0.8\n")
    pl = trg_planner.TRGPlanner(sampler_seed=1)
    pl.setParams(str(tmp_path / "config" / "indoor.yaml"))
    pl.init()
    try:
        assert pl.getFlagPreMap() and not pl.getFlagGraphInit()
        assert pl.getMapEigen("pre").shape[1] == 3
        pl.setPose((1.5, 1.5, 0.0))
        r = pl.processOperation("graph", "expand")
        assert r.success and r.message == "Graph expansion triggered"
        t0 = time.time()
        while not pl.getFlagGraphInit() and time.time() - t0 < 60:
            time.sleep(0.01)
        assert pl.getFlagGraphInit()
        nodes = pl.getTRG().getGraphCopy("global")
        assert len(nodes) > 100 and all(len(n.edges) >= 1 for n in nodes.values())
        # a goal on some far-away node of the graph
        far = max(nodes.values(), key=lambda n: float(np.hypot(n.pos[0] - 1.5, n.pos[1] - 1.5)))
        pl.setGoal(far.pos)
        t0 = time.time()
        while not pl.getFlagPathFound() and time.time() - t0 < 30:
            time.sleep(0.01)
        assert pl.getFlagPathFound()
        path = pl.getPlannedPath("smooth")
        info = pl.getPathInfo()
        assert len(path) > 2 and len(info) == 5 and info[1] > 0
        assert pl.processOperation("graph", "bogus").success is False
    finally:
        pl.shutdown()
