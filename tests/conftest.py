import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (first: a process that uses torch's HIP runtime next to the engine must load
#                torch's bundled ROCm libraries before libtrg_engine.so pulls in the system ones)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "trg-planner_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oa():
    """The CPU oracle wrapper (test infrastructure)."""
    import oracle_api
    oracle_api.build()
    return oracle_api


@pytest.fixture(scope="session")
def synth():
    from trg_planner import synth as s
    return s


@pytest.fixture(scope="session")
def mountain_small(synth):
    # 30 m x 30 m, 90 k points, rough enough to exercise gate / collision / clamp branches
    return synth.mountain_cloud(300, 300, seed=11, amplitude=5.0, wavelength=14.0)


@pytest.fixture(scope="session")
def mountain_gentle(synth):
    return synth.mountain_cloud(300, 300, seed=5)


@pytest.fixture(scope="session")
def indoor_small(synth):
    pts, boxes = synth.indoor_cloud(seed=1, size=(16.0, 12.0), n_boxes=5)
    return synth.voxel_centroids(pts, 0.2)


def weight_report(w_engine, w_oracle, tol=1e-5):
    """Classify per-edge weight differences: (clamp_flips, others_over_tol, max_other).

    A *clamp flip* is an edge where one side is exactly 0 and the other lies in [0.1, 0.1 + tol]:
    the reference's `if (weight < 0.1) weight = 0` cliff (trg.cpp:361-363) decided the other way
    for a weight within tol of 0.1.  Everything else is an ordinary difference."""
    we = np.asarray(w_engine, np.float64)
    wo = np.asarray(w_oracle, np.float64)
    dw = np.abs(we - wo)
    hi = np.maximum(we, wo)
    flip = ((we == 0) != (wo == 0)) & (hi >= 0.1 - tol) & (hi <= 0.1 + tol)
    other = (dw > tol) & ~flip
    return int(flip.sum()), int(other.sum()), float(dw[~flip].max()) if (~flip).any() else 0.0


def assert_graph_equal(g_engine, g_oracle, weight_tol=1e-5, max_clamp_flips=0):
    """Bit-exact structure (ids, CSR, states, xyz, dist); EVERY weight within weight_tol (no
    allowance), except clamp flips, which are counted separately and bounded by max_clamp_flips."""
    assert g_engine.V == g_oracle.V, (g_engine.V, g_oracle.V)
    assert g_engine.E == g_oracle.E, (g_engine.E, g_oracle.E)
    assert np.array_equal(g_engine.rowptr, g_oracle.rowptr)
    assert np.array_equal(g_engine.col, g_oracle.col)
    assert np.array_equal(g_engine.state, g_oracle.state)
    assert np.array_equal(g_engine.xyz.view(np.uint32), g_oracle.xyz.view(np.uint32))
    assert np.array_equal(g_engine.dist.view(np.uint32), g_oracle.dist.view(np.uint32))
    assert np.array_equal(g_engine.cid, g_oracle.cid)
    if g_engine.E:
        flips, others, mx = weight_report(g_engine.w, g_oracle.w, weight_tol)
        assert others == 0, (others, mx)
        assert flips <= max_clamp_flips, (flips, max_clamp_flips)
