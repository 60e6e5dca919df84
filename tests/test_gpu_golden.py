"""GPU: the HIP path against the COMMITTED golden fixtures (tests/golden/trg_golden.npz), which
were produced by the oracle with the reference kdtree.c answering every spatial query."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "trg_golden.npz")
TOL = 1e-5


def _cmp(g, gold, pre):
    assert np.array_equal(g.xyz.view(np.uint32), gold[f"{pre}_xyz"].view(np.uint32))
    for k in ("state", "rowptr", "col", "cid"):
        assert np.array_equal(getattr(g, k), gold[f"{pre}_{k}"]), k
    assert np.array_equal(g.dist.view(np.uint32), gold[f"{pre}_dist"].view(np.uint32))
    assert float(np.abs(g.w - gold[f"{pre}_w"]).max()) <= TOL


def test_mountain_golden(oa, synth):
    import trg_planner
    gold = np.load(GOLD)
    nx, ny, seed = [int(v) for v in gold["m_seed"]]
    cloud = synth.mountain_cloud(nx, ny, seed=seed, amplitude=5.0, wavelength=14.0)
    e = trg_planner.Engine(**oa.MOUNTAIN)
    e.set_sampler(7, 16)
    e.set_option("keep_preclean", 1)
    e.set_global_map(cloud)
    f, c, n = e.is_collision(gold["m_xy"])
    assert np.array_equal(f, gold["m_flag"]) and np.array_equal(c, gold["m_cnt"])
    assert np.array_equal(n, gold["m_n"])
    assert np.array_equal(e.nearest_z(gold["m_xy"]).view(np.uint32), gold["m_z"].view(np.uint32))
    st, npts, w, d = e.edge_risk(gold["m_p1"], gold["m_p2"])
    assert np.array_equal(st, gold["m_status"])
    ok = st == 0
    assert np.array_equal(npts[ok], gold["m_npts"][ok])
    assert np.array_equal(d.view(np.uint32), gold["m_dist"].view(np.uint32))
    assert float(np.abs(w - gold["m_w"]).max()) <= TOL
    e.init_graph([8.0, 8.0, 0.0])
    _cmp(e.graph("preclean"), gold, "m_pre")
    _cmp(e.graph("global"), gold, "m_post")
    for i in range(2):
        path, info = e.plan(gold[f"m_plan{i}_start"], gold[f"m_plan{i}_goal"])
        assert np.array_equal(path, gold[f"m_plan{i}_path"])
        assert info.direct_dist == gold[f"m_plan{i}_info"][0]
        assert info.path_length == gold[f"m_plan{i}_info"][1]
        assert abs(info.avg_risk - gold[f"m_plan{i}_info"][2]) <= TOL
        assert np.array_equal(e.refine_path(path), gold[f"m_plan{i}_smooth"])


def test_indoor_golden(oa):
    import trg_planner
    gold = np.load(GOLD)
    e = trg_planner.Engine(**oa.INDOOR)
    e.set_sampler(5, 16)
    e.set_option("keep_preclean", 1)
    e.set_global_map(gold["i_cloud"])
    e.init_graph([1.5, 1.5, 0.0])
    st = e.stats()  # indoor.yaml turns expandGraph's step 3 on: that runs on the device-resident path too
    assert st["used_device_bfs"] == 1 and st["bfs_fallbacks"] == 0, e.fallback_reason
    _cmp(e.graph("preclean"), gold, "i_pre")
    _cmp(e.graph("global"), gold, "i_post")
