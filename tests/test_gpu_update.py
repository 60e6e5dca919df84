"""GPU: incremental path -- setLocalMap / setLocalGraph / updateGraph /
isFrontier (trg.cpp:195-231, 456-489, 780-803) against the oracle, including the local-map
membership order (container iteration order) and re-expansion of local nodes."""
import numpy as np
import pytest

from conftest import assert_graph_equal

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _obs_crop(cloud, centre, half, box=None):
    m = (np.abs(cloud[:, 0] - centre[0]) < half) & (np.abs(cloud[:, 1] - centre[1]) < half)
    obs = cloud[m].copy()
    if box is not None:  # raise a block of points: an obstacle that was not in the global map
        b = (np.abs(obs[:, 0] - box[0]) < box[2]) & (np.abs(obs[:, 1] - box[1]) < box[2])
        obs[b, 2] += np.float32(1.0) * (np.arange(b.sum()) % 2).astype(np.float32)
    return obs


@pytest.mark.parametrize("replay", ["device", "host"])
def test_update_graph_parity(oa, mountain_gentle, replay):
    import trg_planner
    prm = dict(oa.MOUNTAIN, update_collision_threshold=0.2)
    e = trg_planner.Engine(**prm)
    e.set_sampler(5, 16)
    e.set_option("replay", replay)
    e.set_global_map(mountain_gentle)
    e.init_graph([15.0, 15.0, 0.0])
    o = oa.Oracle(**prm)
    o.set_sampler(5, 0, 16)
    o.set_global_map(mountain_gentle)
    assert o.init_graph([15.0, 15.0, 0.0])
    assert_graph_equal(e.graph("global"), o.graph(0), TOL)
    # the oracle's second witness (covariance accumulated in fp64, as the engine does) goes through the same
    # updates: its weights are the engine's, bit for bit
    o2 = oa.Oracle(**prm)
    o2.set_sampler(5, 0, 16)
    o2.set_cov_f64(True)
    o2.set_global_map(mountain_gentle)
    assert o2.init_graph([15.0, 15.0, 0.0])

    poses = [(12.0, 12.0), (13.0, 12.5), (14.0, 13.0)]
    for k, pose in enumerate(poses):
        obs = _obs_crop(mountain_gentle, pose, 4.0, box=(pose[0] + 2.0, pose[1] + 1.0, 0.6))
        e.set_local_map(pose, obs)
        o.set_local_map(pose, obs)
        o2.set_local_map(pose, obs)
        # isFrontier / isCollision on the local map agree before the update
        loc = e.graph("local")
        assert loc.V > 10
        fe = e.is_frontier(loc.xyz[:, :2])
        fo = o.is_frontier(loc.xyz[:, :2])
        assert np.array_equal(fe, fo)
        ce, _, ne = e.is_collision(loc.xyz[:, :2], kind="local", threshold=prm["update_collision_threshold"])
        co, _, no = o.is_collision(loc.xyz[:, :2], 1, prm["update_collision_threshold"])
        assert np.array_equal(ce, co) and np.array_equal(ne, no)
        e.update_graph()
        o.update_graph()
        o2.update_graph()
        ge, go = e.graph("global"), o.graph(0)
        g2 = o2.graph(0)
        assert np.array_equal(g2.col, ge.col)
        assert np.array_equal(ge.w.view(np.uint32), g2.w.view(np.uint32)), (k, float(np.abs(ge.w - g2.w).max()))
        assert ge.V == go.V and ge.E == go.E, (k, ge.V, go.V, ge.E, go.E)
        assert np.array_equal(ge.rowptr, go.rowptr) and np.array_equal(ge.col, go.col)
        assert np.array_equal(ge.state, go.state)
        assert np.array_equal(ge.xyz.view(np.uint32), go.xyz.view(np.uint32))
        assert np.array_equal(ge.dist.view(np.uint32), go.dist.view(np.uint32))
        assert float(np.abs(ge.w - go.w).max()) <= TOL
    # the obstacle really invalidated / removed something and frontier states exist
    assert (go.state == 1).any()
    # planning still agrees after updates
    pe, ie = e.plan((10.0, 10.0), (20.0, 19.0, 0.0))
    po, io = o.plan((10.0, 10.0), (20.0, 19.0, 0.0))
    assert np.array_equal(pe, po) and ie.path_length == io[1]
