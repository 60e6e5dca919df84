"""GPU: BASELINE-scale checks through size-independent properties (the CPU oracle would need
minutes at these sizes): the two independent replay implementations of the engine -- the
device-resident BFS and the sequential host replay -- must agree bit for bit on a 1 M-point map
(BASELINE config 2 size), and the graph must satisfy the reference's invariants."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _invariants(g, expand_dist):
    assert (g.state != -1).all() and (np.diff(g.rowptr) >= 1).all()       # cleanGraph post-condition
    assert (g.dist < 2.5 * expand_dist).all()                              # trg.cpp:279
    nz = g.w[g.w != 0]
    assert ((nz >= 0.1) & (nz <= 0.4761)).all()                            # trg.cpp:359-363
    src = np.repeat(np.arange(g.V, dtype=np.int64), np.diff(g.rowptr))
    key = src * g.V + g.col
    rev = g.col.astype(np.int64) * g.V + src
    assert np.array_equal(np.sort(key), np.sort(rev))                      # edges are symmetric
    assert np.unique(key).size == key.size                                 # wireEdge's dedupe
    # nodes are at least robot_size apart only in a statistical sense (merge test is against the
    # NEAREST node); what must hold exactly: every edge length equals the fp32 node distance
    d = np.sqrt(((g.xyz[src, 0] - g.xyz[g.col, 0]) ** 2 + (g.xyz[src, 1] - g.xyz[g.col, 1]) ** 2))
    assert np.abs(d - g.dist).max() < 1e-5


def test_c2_size_against_the_live_oracle(oa, synth):
    """BASELINE config 2 at its real size (1.0 M points, mountain.yaml, S = 7) against the CPU oracle run
    right here (about 5 s of host time; its spatial queries through the reference kdtree.c when
    oracle/_ref is there): structure bit-exact, every weight within 1e-5, and the five planning queries."""
    import trg_planner
    from conftest import assert_graph_equal
    cloud = synth.mountain_tile(0, 1000, 0, 1000, seed=20250418)
    prm = dict(expand_dist=0.6, robot_size=0.3, sample_num=7, height_threshold=0.16,
               collision_threshold=0.1, update_collision_threshold=0.5, safety_factor=3.0,
               goal_tolerance=0.8)
    e = trg_planner.Engine(**prm)
    e.set_sampler(7, 16)
    e.set_global_map(cloud)
    e.init_graph([50.0, 50.0, 0.0])
    assert e.stats()["used_device_bfs"] == 1, e.fallback_reason
    oa.use_reference_kd(True)
    try:
        o = oa.Oracle(**prm)
        o.set_sampler(7, 0, 16)
        o.set_global_map(cloud)
        assert o.init_graph([50.0, 50.0, 0.0])
        g, r = e.graph("global"), o.graph(0)
        assert g.V > 40000
        assert_graph_equal(g, r, 1e-5)
        c, st = o.counters(), e.stats()
        assert st["trials"] == c["trials"] and st["samples"] == c["samples"]
        assert st["created_nodes"] == c["created"] and st["invalid_nodes"] == c["invalid_created"]
        rng = np.random.default_rng(1)
        for _ in range(5):  # start/goal pairs in the style of the reference's run_trg_planner.py:35-43
            s = rng.uniform(10, 90, 2).astype(np.float32)
            goal = np.append(rng.uniform(10, 90, 2), 0.0).astype(np.float32)
            pe, ie = e.plan(s, goal)
            po, io = o.plan(s, goal)
            assert np.array_equal(pe.view(np.uint32), po.view(np.uint32))
            assert ie.path_length == io[1]
    finally:
        oa.use_reference_kd(False)


def test_c1_indoor_standin_against_the_live_oracle(oa, synth):
    """BASELINE config 1 as bench.py --workload c1 runs it: the 40 m x 30 m indoor stand-in, voxel filter 0.2 on
    the GPU, indoor.yaml parameters (expandGraph's step 3 ON, trg.cpp:429) -- on the device-resident path, against
    the CPU oracle run right here (under a second), and against the engine's own sequential host replay."""
    import trg_planner
    from conftest import assert_graph_equal
    raw, _ = synth.indoor_cloud(seed=1, size=(40.0, 30.0))
    prm = dict(oa.INDOOR)
    pre = trg_planner.Engine(**prm)
    cloud = pre.voxel_filter(raw, 0.2)
    pre.close()
    start = [3.27, 4.12, 0.0]
    graphs = {}
    for mode in ("device", "host"):
        e = trg_planner.Engine(**prm)
        e.set_sampler(7, 16)
        e.set_option("keep_preclean", 1)
        e.set_option("replay", mode)
        e.set_global_map(cloud)
        e.init_graph(start)
        st = e.stats()
        assert st["used_device_bfs"] == (1 if mode == "device" else 0) and st["bfs_fallbacks"] == 0, e.fallback_reason
        graphs[mode] = (e.graph("preclean"), e.graph("global"), st)
    o = oa.Oracle(**prm)
    o.set_sampler(7, 0, 16)
    o.set_global_map(cloud)
    assert o.init_graph(start)
    assert o.graph(0).V > 5000
    for mode in ("device", "host"):
        assert_graph_equal(graphs[mode][0], o.graph(1), 1e-5)
        assert_graph_equal(graphs[mode][1], o.graph(0), 1e-5)
    c = o.counters()
    st = graphs["device"][2]
    assert st["trials"] == c["trials"] and st["samples"] == c["samples"]
    assert st["created_nodes"] == c["created"] and st["invalid_nodes"] == c["invalid_created"]
    assert c["invalid_created"] > 0  # (nodes whose every call failed exist in this map: the rescue logic had work)


def test_c2_size_device_and_host_replay_agree(synth):
    import trg_planner
    cloud = synth.mountain_tile(0, 1000, 0, 1000, seed=20250418)   # 1.0 M points, 100 m x 100 m
    prm = dict(expand_dist=0.6, robot_size=0.3, sample_num=7, height_threshold=0.16,
               collision_threshold=0.1, update_collision_threshold=0.5, safety_factor=3.0,
               goal_tolerance=0.8)
    graphs = {}
    for mode in ("device", "host"):
        e = trg_planner.Engine(**prm)
        e.set_sampler(7, 16)
        e.set_option("replay", mode)
        e.set_global_map(cloud)
        e.init_graph([50.0, 50.0, 0.0])
        st = e.stats()
        assert st["used_device_bfs"] == (1 if mode == "device" else 0), e.fallback_reason
        graphs[mode] = (e.graph("global"), st)
    gd, sd = graphs["device"]
    gh, sh = graphs["host"]
    assert gd.V > 40000 and gd.V == gh.V and gd.E == gh.E
    for k in ("rowptr", "col", "state", "cid"):
        assert np.array_equal(getattr(gd, k), getattr(gh, k)), k
    assert np.array_equal(gd.xyz.view(np.uint32), gh.xyz.view(np.uint32))
    assert np.array_equal(gd.dist.view(np.uint32), gh.dist.view(np.uint32))
    assert np.array_equal(gd.w.view(np.uint32), gh.w.view(np.uint32))       # same kernels: bitwise
    for k in ("expanded_nodes", "trials", "samples", "created_nodes", "invalid_nodes"):
        assert sd[k] == sh[k], k
    _invariants(gd, prm["expand_dist"])
    # five start/goal pairs in the style of the reference's run_trg_planner.py:35-43
    e = trg_planner.Engine(**prm)
    e.set_sampler(7, 16)
    e.set_global_map(cloud)
    e.init_graph([50.0, 50.0, 0.0])
    rng = np.random.default_rng(1)
    found = 0
    for _ in range(5):
        s = rng.uniform(10, 90, 2).astype(np.float32)
        g = np.append(rng.uniform(10, 90, 2), 0.0).astype(np.float32)
        path, info = e.plan(s, g)
        if info.num_points:
            found += 1
            assert info.path_length >= info.direct_dist * 0.999
    assert found >= 3
