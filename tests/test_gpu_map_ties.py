"""GPU: nearest-MAP-point ties.  addNode takes the z of kd_nearest's result (trg.cpp:244-247); when
several map points are at exactly the same fp32 distance the reference returns whichever its
insertion-built map tree visits first (kdtree.c:303-362).  The engine never builds that tree; it
reproduces the visiting order through the lowest-common-ancestor argument documented at
map_nn_exact (trg_engine.cpp).  These tests construct clouds where such ties are certain."""
import numpy as np
import pytest

from conftest import assert_graph_equal

pytestmark = pytest.mark.gpu


def _lattice(n=48, step=0.125, seed=3):
    """n x n lattice with power-of-two spacing (midpoints are exact in fp32), random z, shuffled
    insertion order -- the tree shape, and with it the tie winner, depends on that order."""
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    xyz = np.stack([ii.ravel() * step, jj.ravel() * step, rng.uniform(0, 1, n * n)], 1).astype(np.float32)
    return xyz[rng.permutation(n * n)]


@pytest.mark.parametrize("seed,n", [(3, 48), (4, 48), (5, 144)])
def test_nearest_z_tie_follows_map_tree_order(oa, seed, n):
    """(n = 48: the whole insertion tree is inside the host-side top of the tree; n = 144, 20 736
    points: the walk starts on the host and continues on the device from the region below the top.)"""
    import trg_planner
    cloud = _lattice(n=n, seed=seed)
    step = 0.125
    rng = np.random.default_rng(seed + 100)
    # edge midpoints (2 points tied), cell centres (4 tied), and lattice-point reflections
    q = []
    for _ in range(60):
        i, j = rng.integers(1, n - 2, 2)
        q.append([(i + 0.5) * step, j * step])
        q.append([i * step, (j + 0.5) * step])
        q.append([(i + 0.5) * step, (j + 0.5) * step])
    q = np.asarray(q, np.float32)
    prm = dict(oa.MOUNTAIN)
    e = trg_planner.Engine(**prm)
    e.set_global_map(cloud)
    o = oa.Oracle(**prm)
    o.set_global_map(cloud)
    ze = e.nearest_z(q)
    zo = o.nearest_z(q)
    st = e.stats()
    assert st["map_nn_resolved"] == len(q) and st["map_nn_unresolved"] == 0
    assert np.array_equal(np.asarray(ze, np.float32).view(np.uint32), zo.view(np.uint32))
    # the lowest cloud index alone would have been wrong for a good share of them
    order = {tuple(np.round(p[:2] / step).astype(int)): k for k, p in enumerate(cloud)}
    naive_wrong = 0
    for p, z in zip(q, zo):
        fx, fy = p[0] / step, p[1] / step
        cands = {(int(np.floor(fx)), int(np.floor(fy))), (int(np.ceil(fx)), int(np.floor(fy))),
                 (int(np.floor(fx)), int(np.ceil(fy))), (int(np.ceil(fx)), int(np.ceil(fy)))}
        k = min(order[c] for c in cands)
        naive_wrong += int(cloud[k, 2] != z)
    assert naive_wrong > 10


def _cloud_with_duplicates(synth, frac=0.03, seed=5):
    base = synth.mountain_cloud(300, 300, seed=seed)
    rng = np.random.default_rng(seed)
    pick = rng.choice(base.shape[0], int(frac * base.shape[0]), replace=False)
    dup = base[pick].copy()
    dup[:, 2] += rng.uniform(0.01, 0.04, dup.shape[0]).astype(np.float32)   # same x, y; other z
    cloud = np.concatenate([base, dup])
    return cloud[rng.permutation(cloud.shape[0])]


@pytest.mark.parametrize("replay", ["device", "host"])
def test_build_with_tied_elevation_lookups(oa, synth, replay):
    """3 % of the map points have a twin at the same (x, y) with another z: every sample whose
    nearest point is one of them ties, and the node's z -- hence the slope gate of its edges --
    depends on the tree's visiting order."""
    import trg_planner
    cloud = _cloud_with_duplicates(synth)
    prm = dict(oa.MOUNTAIN)
    start = [15.0, 15.0, 0.0]
    e = trg_planner.Engine(**prm)
    e.set_sampler(7, 16)
    e.set_option("keep_preclean", 1)
    e.set_option("replay", replay)
    e.set_global_map(cloud)
    e.init_graph(start)
    o = oa.Oracle(**prm)
    o.set_sampler(7, 0, 16)
    o.set_global_map(cloud)
    assert o.init_graph(start)
    st = e.stats()
    assert st["used_device_bfs"] == (1 if replay == "device" else 0), e.fallback_reason
    assert st["map_nn_resolved"] > 20 and st["map_nn_unresolved"] == 0, st
    go = o.graph(1)
    assert go.V > 200
    assert_graph_equal(e.graph("preclean"), go, 1e-5)
    assert_graph_equal(e.graph("global"), o.graph(0), 1e-5)
