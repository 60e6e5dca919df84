"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on seeded inputs.

Bars: bit-exact for every index / integer / fp32-decision quantity (ids, CSR, states, node xyz,
edge dist, hit counts, status codes); |weight difference| <= 1e-5 (north_star tolerance) for the
fp32 PCA risk weight.
"""
import numpy as np
import pytest

from conftest import assert_graph_equal

pytestmark = pytest.mark.gpu

WEIGHT_TOL = 1e-5  # BASELINE.json north_star: "within 1e-5 for float risk"
MOUNTAIN_S16 = dict(expand_dist=0.6, robot_size=0.3, height_threshold=0.16, collision_threshold=0.1,
                    update_collision_threshold=0.5, safety_factor=3.0, goal_tolerance=0.8, sample_num=16)


def _engine(params, **kw):
    import trg_planner
    return trg_planner.Engine(**params, **kw)


def _probes(cloud, m, seed, margin=0.5):
    rng = np.random.default_rng(seed)
    lo = cloud[:, :2].min(0) - margin
    hi = cloud[:, :2].max(0) + margin
    return rng.uniform(lo, hi, size=(m, 2)).astype(np.float32)


def test_library_loaded_and_arch():
    import trg_planner
    e = trg_planner.Engine()
    assert e.arch.startswith("gfx950")


def test_map_index_is_sorted_permutation(mountain_small):
    e = _engine({})
    e.set_global_map(mountain_small)
    n = mountain_small.shape[0]
    x, y, z, perm, wh, org = e.map_index("global", n)
    assert np.array_equal(np.sort(perm), np.arange(n))
    assert np.array_equal(x, mountain_small[perm, 0])
    assert np.array_equal(y, mountain_small[perm, 1])
    assert np.array_equal(z, mountain_small[perm, 2])
    g = org[2]
    cx = np.clip(np.floor((x - org[0]) * np.float32(1.0 / g)), 0, wh[0] - 1).astype(np.int64)
    cy = np.clip(np.floor((y - org[1]) * np.float32(1.0 / g)), 0, wh[1] - 1).astype(np.int64)
    cell = cy * wh[0] + cx
    assert np.all(np.diff(cell) >= 0)
    same = np.diff(cell) == 0
    assert np.all(np.diff(perm)[same] > 0)  # deterministic order inside a cell


def test_map_index_through_bins_equals_the_direct_build(mountain_small, monkeypatch):
    """The index built through bins (the default) and the one built with global atomics (huge grids only,
    forced here through TRG_INDEX_DIRECT) are the same arrays: the order inside a cell is by original index
    in both.  A shuffled cloud (as the benchmark's) and one in scan order."""
    rng = np.random.default_rng(5)
    for cloud in (mountain_small[rng.permutation(mountain_small.shape[0])], np.ascontiguousarray(mountain_small)):
        n = cloud.shape[0]
        monkeypatch.delenv("TRG_INDEX_DIRECT", raising=False)
        e1 = _engine({})
        e1.set_global_map(cloud)
        a = e1.map_index("global", n)
        monkeypatch.setenv("TRG_INDEX_DIRECT", "1")
        e2 = _engine({})
        e2.set_global_map(cloud)
        b = e2.map_index("global", n)
        monkeypatch.delenv("TRG_INDEX_DIRECT", raising=False)
        for u, v in zip(a[:4], b[:4]):
            assert np.array_equal(u, v)
        assert tuple(a[4]) == tuple(b[4]) and tuple(a[5]) == tuple(b[5])


def test_sampler_table_matches_oracle(oa):
    e = _engine({})
    o = oa.Oracle()
    for bits in (12, 16):
        e.set_sampler(3, bits)
        e.L.trg_engine_get_sampler_table  # noqa: B018
        # the engine builds its table lazily for the configured sampler
        import ctypes as C
        from trg_planner._engine import TrgSampler
        e.sampler = TrgSampler(3, bits)
        e.set_global_map(np.zeros((4, 3), np.float32) + np.arange(4, dtype=np.float32)[:, None])
        try:
            e.init_graph([0, 0, 0])
        except Exception:
            pass
        ce, se = e.sampler_table()
        o.set_sampler(3, 0, bits)
        co, so = o.table()
        assert np.array_equal(ce.view(np.uint32), co.view(np.uint32))
        assert np.array_equal(se.view(np.uint32), so.view(np.uint32))


@pytest.mark.parametrize("cloud_name", ["mountain_small", "indoor_small"])
def test_is_collision_parity(oa, request, cloud_name):
    cloud = request.getfixturevalue(cloud_name)
    prm = oa.MOUNTAIN if cloud_name.startswith("mountain") else oa.INDOOR
    e = _engine(prm)
    e.set_global_map(cloud)
    o = oa.Oracle(**prm)
    o.set_global_map(cloud)
    xy = _probes(cloud, 4096, 1)
    fe, ce, ne = e.is_collision(xy, threshold=prm["collision_threshold"])
    fo, co, no = o.is_collision(xy, 0, prm["collision_threshold"])
    assert np.array_equal(ne, no)
    assert np.array_equal(ce, co)
    assert np.array_equal(fe, fo)
    assert (no == 0).any() and (fo == 0).any() and (fo == 1).any()  # empty discs, free, blocked


def test_nearest_z_parity(oa, mountain_small):
    e = _engine(oa.MOUNTAIN)
    e.set_global_map(mountain_small)
    o = oa.Oracle(**oa.MOUNTAIN)
    o.set_global_map(mountain_small)
    xy = _probes(mountain_small, 4096, 2, margin=3.0)  # includes far-outside probes
    ze = e.nearest_z(xy)
    zo = o.nearest_z(xy)
    assert np.array_equal(ze.view(np.uint32), zo.view(np.uint32))


def _edge_pairs(cloud, oracle, m, seed, dmax):
    rng = np.random.default_rng(seed)
    lo = cloud[:, :2].min(0) + 1.0
    hi = cloud[:, :2].max(0) - 1.0
    a = rng.uniform(lo, hi, size=(m, 2)).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, m)
    d = rng.uniform(0.05, dmax, m)
    b = (a + np.stack([d * np.cos(ang), d * np.sin(ang)], 1)).astype(np.float32)
    za = oracle.nearest_z(a)
    zb = oracle.nearest_z(b)
    return np.concatenate([a, za[:, None]], 1), np.concatenate([b, zb[:, None]], 1)


@pytest.mark.parametrize("cloud_name", ["mountain_small", "mountain_gentle", "indoor_small"])
def test_edge_risk_parity(oa, request, cloud_name):
    cloud = request.getfixturevalue(cloud_name)
    prm = oa.MOUNTAIN if cloud_name.startswith("mountain") else oa.INDOOR
    e = _engine(prm)
    e.set_global_map(cloud)
    o = oa.Oracle(**prm)
    o.set_global_map(cloud)
    p1, p2 = _edge_pairs(cloud, o, 3000, 3, 2.4 * prm["expand_dist"])
    se, ne, we, de = e.edge_risk(p1, p2)
    so, no, wo, do = o.edge_risk(p1, p2)
    assert np.array_equal(de.view(np.uint32), do.view(np.uint32))
    assert np.array_equal(se, so)
    ok = so == 0
    assert np.array_equal(ne[ok | (so == 4)], no[ok | (so == 4)])
    assert ok.sum() > 100
    # Two witnesses.  (i) The literal fp32 restatement of trg.cpp:332-338: every edge within 1e-5,
    # clamp flips counted apart.  (ii) The same formula with fp64 accumulation: far tighter.
    from conftest import weight_report
    flips, others, mx = weight_report(we[ok], wo[ok], WEIGHT_TOL)
    assert others == 0 and flips == 0, (flips, others, mx)
    o.set_cov_f64(True)
    so2, _, wo2, _ = o.edge_risk(p1, p2)
    assert np.array_equal(so2, so)
    flips2, others2, mx2 = weight_report(we[ok], wo2[ok], 2e-6)
    assert others2 == 0 and flips2 == 0, (flips2, others2, mx2)


def _build_both(oa, prm, cloud, start, seed, replay=None):
    e = _engine(prm)
    e.set_sampler(seed, 16)
    e.set_option("keep_preclean", 1)
    if replay:
        e.set_option("replay", replay)
    e.set_global_map(cloud)
    e.init_graph(start)
    o = oa.Oracle(**prm)
    o.set_sampler(seed, 0, 16)
    o.set_global_map(cloud)
    assert o.init_graph(start)
    return e, o


@pytest.mark.parametrize("replay", ["device", "host"])
@pytest.mark.parametrize("case", ["mountain_rough_S7", "mountain_gentle_S16", "indoor_S15"])
def test_init_graph_parity(oa, request, case, replay):
    if case == "mountain_rough_S7":
        cloud = request.getfixturevalue("mountain_small")
        prm = dict(oa.MOUNTAIN)
        start = [15.0, 15.0, 0.0]
    elif case == "mountain_gentle_S16":
        cloud = request.getfixturevalue("mountain_gentle")
        prm = dict(oa.MOUNTAIN, sample_num=16)
        start = [15.0, 15.0, 0.0]
    else:
        cloud = request.getfixturevalue("indoor_small")
        prm = dict(oa.INDOOR)
        start = [1.5, 1.5, 0.0]
    e, o = _build_both(oa, prm, cloud, start, seed=7, replay=replay)
    used_device = e.stats()["used_device_bfs"]
    # the device-resident BFS serves both kinds of configuration: expandGraph's step 3 (trg.cpp:429) off
    # (mountain.yaml) and on (indoor.yaml: rescue search inside the level, neighbour calls after the loop)
    assert used_device == (1 if replay == "device" else 0), e.fallback_reason
    assert e.stats()["bfs_fallbacks"] == 0, e.fallback_reason
    pre_e, pre_o = e.graph("preclean"), o.graph(1)
    assert pre_o.V > 200, pre_o.V
    assert_graph_equal(pre_e, pre_o, WEIGHT_TOL)
    ge, go = e.graph("global"), o.graph(0)
    assert_graph_equal(ge, go, WEIGHT_TOL)
    # the oracle's second witness (covariance accumulated in fp64, as the engine does): the same floats
    o2 = oa.Oracle(**prm)
    o2.set_sampler(7, 0, 16)
    o2.set_cov_f64(True)
    o2.set_global_map(cloud)
    assert o2.init_graph(start)
    g2 = o2.graph(0)
    assert np.array_equal(g2.col, ge.col)
    assert np.array_equal(ge.w.view(np.uint32), g2.w.view(np.uint32)), float(np.abs(ge.w - g2.w).max())
    st = e.stats()
    c = o.counters()
    assert st["expanded_nodes"] == c["expanded"]
    assert st["trials"] == c["trials"]
    assert st["samples"] == c["samples"]
    assert st["created_nodes"] == c["created"]
    assert st["invalid_nodes"] == c["invalid_created"]
    # algorithmic bytes of the sampling kernel == 12 B x the map points inside the discs of the
    # reference's own sampling loop (B_alg counted by the oracle's instrumentation)
    assert st["bytes_sample_kernel"] == 12 * c["sample_hits"], (st["bytes_sample_kernel"], c["sample_hits"])
    if used_device:
        # parent edges of the created nodes: the very wireEdge(node, new_node) calls of trg.cpp:425
        assert st["bytes_spec_created"] == 12 * c["wire_hits_new"], (st["bytes_spec_created"], c["wire_hits_new"])
        assert st["bytes_spec_kernel"] >= st["bytes_spec_created"]      # + speculation on merged candidates
        # deferred edges: never fewer queries than the reference's other wireEdge calls; the excess is
        # the second selection round (all remaining calls of a pair whose first call failed)
        assert st["bytes_edge_kernel"] >= 12 * c["wire_hits_other"]
        assert st["bytes_edge_kernel"] <= 12 * c["wire_hits_other"] * 1.02 + 12 * 2000
    assert st["nn_ties"] == 0 and st["map_nn_ties"] == 0
    # invariants of the reference
    assert (ge.state != -1).all() and (np.diff(ge.rowptr) >= 1).all()
    assert (ge.dist < 2.5 * prm["expand_dist"]).all()
    nz = ge.w[ge.w != 0]
    assert ((nz >= 0.1) & (nz <= 0.4761)).all()


def test_plan_parity(oa, mountain_gentle):
    prm = dict(oa.MOUNTAIN)
    e, o = _build_both(oa, prm, mountain_gentle, [15.0, 15.0, 0.0], seed=3)
    rng = np.random.default_rng(0)
    for _ in range(5):
        s = rng.uniform(3, 27, 2).astype(np.float32)
        g = np.append(rng.uniform(3, 27, 2), 0.0).astype(np.float32)
        pe, ie = e.plan(s, g)
        po, io = o.plan(s, g)
        assert pe.shape == po.shape and pe.shape[0] > 1
        assert np.array_equal(pe.view(np.uint32), po.view(np.uint32))
        assert ie.direct_dist == io[0] and ie.path_length == io[1]
        assert abs(ie.avg_risk - io[2]) <= WEIGHT_TOL
        re_, ro = e.refine_path(pe), oa.Oracle.refine(po)
        assert np.array_equal(re_.view(np.uint32), ro.view(np.uint32))


def test_plan_on_csr_many_queries(oa, mountain_gentle):
    """planSafePath runs straight on the CSR; start / goal nodes come from the node grid plus the tree-order
    arguments (no node tree is built).  200 queries against the oracle, including goals between two or
    three nodes (several hits within robot_size: the reference takes the one its range walk reaches
    last), goals exactly on nodes, and goals off the graph (no hit: nearest node in node-map order)."""
    prm = dict(oa.MOUNTAIN)
    e, o = _build_both(oa, prm, mountain_gentle, [15.0, 15.0, 0.0], seed=3)
    g = e.graph("global")
    rng = np.random.default_rng(11)
    goals = []
    for _ in range(60):   # midpoints of edges: both ends are usually within robot_size... or not
        a = int(rng.integers(0, g.V))
        if g.rowptr[a + 1] == g.rowptr[a]:
            continue
        b = int(g.col[g.rowptr[a] + rng.integers(0, g.rowptr[a + 1] - g.rowptr[a])])
        goals.append((g.xyz[a] + g.xyz[b]) * np.float32(0.5))
    for _ in range(40):   # centroids of a node and two of its neighbours
        a = int(rng.integers(0, g.V))
        nb = g.col[g.rowptr[a]:g.rowptr[a + 1]]
        if len(nb) >= 2:
            goals.append((g.xyz[a] + g.xyz[nb[0]] + g.xyz[nb[1]]) / np.float32(3.0))
    for _ in range(30):   # exactly on a node
        goals.append(g.xyz[int(rng.integers(0, g.V))].copy())
    for _ in range(40):   # anywhere, also outside the map
        goals.append(np.append(rng.uniform(-5, 35, 2), 0.0).astype(np.float32))
    multi = 0
    for k, goal in enumerate(goals):
        goal = np.asarray(goal, np.float32)
        d = np.hypot(g.xyz[:, 0] - goal[0], g.xyz[:, 1] - goal[1])
        multi += int((d < prm["robot_size"]).sum() >= 2)
        s = rng.uniform(3, 27, 2).astype(np.float32)
        pe, ie = e.plan(s, goal)
        po, io = o.plan(s, goal)
        assert pe.shape == po.shape, (k, pe.shape, po.shape)
        assert np.array_equal(pe.view(np.uint32), po.view(np.uint32)), k
        if pe.shape[0]:
            assert ie.direct_dist == io[0] and ie.path_length == io[1]
            assert abs(ie.avg_risk - io[2]) <= WEIGHT_TOL
    assert multi >= 20, multi  # the several-hits branch was really exercised
    assert e.stats()["used_device_bfs"] == 1


def test_plan_batch_equals_consecutive_plans(oa, mountain_gentle):
    """trg_engine_plan_batch = m planSafePath calls in one boundary crossing (the loop over
    start/goal pairs of the reference's run_trg_planner.py:35-43), including a query without a path
    and a truncated output buffer."""
    prm = dict(oa.MOUNTAIN)
    e, o = _build_both(oa, prm, mountain_gentle, [15.0, 15.0, 0.0], seed=3)
    rng = np.random.default_rng(4)
    starts = rng.uniform(3, 27, (8, 2)).astype(np.float32)
    goals = np.concatenate([rng.uniform(3, 27, (8, 2)), np.zeros((8, 1))], 1).astype(np.float32)
    single = [e.plan(s, g) for s, g in zip(starts, goals)]
    batch = e.plan_batch(starts, goals)
    assert len(batch) == 8
    for (ps, is_), (pb, ib) in zip(single, batch):
        assert np.array_equal(ps.view(np.uint32), pb.view(np.uint32))
        assert (is_.num_points, is_.direct_dist, is_.path_length, is_.avg_risk) == \
               (ib.num_points, ib.direct_dist, ib.path_length, ib.avg_risk)
    for (pb, ib), (s, g) in zip(batch, zip(starts, goals)):
        po, io = o.plan(s, g)
        assert np.array_equal(pb.view(np.uint32), po.view(np.uint32))
    # a buffer that only holds the first path and a half: later ranges are cut, lengths are kept
    cap = batch[0][0].shape[0] + batch[1][0].shape[0] // 2
    cut = e.plan_batch(starts, goals, path_cap=cap)
    assert cut[0][0].shape == batch[0][0].shape and cut[1][0].shape[0] == cap - batch[0][0].shape[0]
    assert cut[2][0].shape[0] == 0 and cut[2][1].num_points == batch[2][1].num_points


def test_repeated_builds_are_identical(oa, mountain_small):
    """Same engine, same inputs, twice: the graph (including the container-order renumbering of
    cleanGraph, which depends on the hash table's bucket history) must match an oracle that went
    through the same two calls."""
    prm = dict(oa.MOUNTAIN)
    e = _engine(prm)
    e.set_sampler(11, 16)
    e.set_global_map(mountain_small)
    o = oa.Oracle(**prm)
    o.set_sampler(11, 0, 16)
    o.set_global_map(mountain_small)
    for start in ([15.0, 15.0, 0.0], [9.0, 21.0, 0.0]):
        e.init_graph(start)
        assert o.init_graph(start)
        assert_graph_equal(e.graph("global"), o.graph(0), WEIGHT_TOL)


def test_tie_levels_replayed_on_host_give_the_same_graph(oa, mountain_small):
    """An exact fp32 nearest-node tie makes the device hand ONE BFS level to the host (where the
    reference's kd-tree order is reproduced) and then continue.  Real ties are ~1e-7 per query, so
    the path is forced here: every 3rd level is treated as tie-affected, alternately before and
    after its commit (which exercises the undo).  The graph must still equal the oracle's."""
    prm = dict(oa.MOUNTAIN, sample_num=10)
    e = _engine(prm)
    e.set_sampler(21, 16)
    e.set_option("keep_preclean", 1)
    e.set_option("debug_tie_every", 3)
    e.set_global_map(mountain_small)
    e.init_graph([15.0, 15.0, 0.0])
    st = e.stats()
    assert st["used_device_bfs"] == 1 and st["bfs_host_levels"] >= 5, st
    o = oa.Oracle(**prm)
    o.set_sampler(21, 0, 16)
    o.set_global_map(mountain_small)
    assert o.init_graph([15.0, 15.0, 0.0])
    assert_graph_equal(e.graph("preclean"), o.graph(1), WEIGHT_TOL)
    assert_graph_equal(e.graph("global"), o.graph(0), WEIGHT_TOL)
    c = o.counters()
    assert st["trials"] == c["trials"] and st["samples"] == c["samples"]
    assert st["created_nodes"] == c["created"] and st["invalid_nodes"] == c["invalid_created"]


@pytest.mark.parametrize("bits", [2, 4, 5])
@pytest.mark.parametrize("inplace", [1, 0])
def test_natural_node_ties_settled_like_the_reference(oa, synth, bits, inplace):
    """Real nearest-node ties, hundreds of them: with only 4 / 16 / 32 sampling directions the nodes
    fall on a few lattices and samples sit at exactly the same fp32 distance from two of them.  The
    reference's answer is its kd-tree's visiting order.  Default: the committed level stands and
    the host decides just the tied slots (tie_inplace) or hands k_level_resolve the winner among
    pre-level nodes (both counted in bfs_tie_fixups); tie_inplace=0, levels with more than
    BFS_TIE_CAP tied slots, and ties that put the level's node set in doubt replay the level on the
    host.  Every mode must equal the oracle."""
    cloud = synth.mountain_cloud(260, 260, seed=5, amplitude=0.3)
    prm = dict(oa.MOUNTAIN, sample_num=8)
    start = [13.0, 13.0, 0.0]
    e = _engine(prm)
    e.set_sampler(33, bits)
    e.set_option("keep_preclean", 1)
    e.set_option("tie_inplace", inplace)
    e.set_global_map(cloud)
    e.init_graph(start)
    st = e.stats()
    assert st["used_device_bfs"] == 1 and st["bfs_fallbacks"] == 0, (st, e.fallback_reason)
    assert st["nn_ties"] > 0, st
    assert st["bfs_tie_fixups"] > 0, st  # (ties among pre-level nodes are always fixed in place)
    if not inplace and bits > 2:
        assert st["bfs_host_levels"] > 0, st
    o = oa.Oracle(**prm)
    o.set_sampler(33, 0, bits)
    o.set_global_map(cloud)
    assert o.init_graph(start)
    assert_graph_equal(e.graph("preclean"), o.graph(1), WEIGHT_TOL)
    assert_graph_equal(e.graph("global"), o.graph(0), WEIGHT_TOL)
    c = o.counters()
    assert st["trials"] == c["trials"] and st["samples"] == c["samples"]
    assert st["created_nodes"] == c["created"] and st["invalid_nodes"] == c["invalid_created"]


@pytest.mark.parametrize("rerun_fails", [False, True], ids=["ticketed-repeat", "host-replay"])
def test_stalled_resolve_is_undone_and_repeated(oa, mountain_small, rerun_fails):
    """k_level_resolve's inter-workgroup wait is bounded; when it runs out (BFS_ERR_STALL) the level is
    left PARTIALLY decided -- outcomes 0 / 7 in c_outcome, commit and emit have run on them.  The
    hook leaves the middle candidate of one mid-build level undecided (everything that waits for it
    runs into the bound): the engine must take the level back (k_bfs_undo_slots) and repeat the launch
    with start tickets as workgroup indices (waits are then for running workgroups by construction);
    if that repeat stalls as well (debug_wait_rerun) the level is replayed on the host.  Either way the
    device goes on and the graph must still equal the oracle's."""
    prm = dict(oa.MOUNTAIN, sample_num=10)
    e0 = _engine(prm)
    e0.set_sampler(21, 16)
    e0.set_global_map(mountain_small)
    e0.init_graph([15.0, 15.0, 0.0])
    natural = e0.stats()["bfs_host_levels"]  # this cloud has one real nearest-node tie (level 71)
    assert e0.stats()["bfs_ticket_reruns"] == 0
    e0.close()
    e = _engine(prm)
    e.set_sampler(21, 16)
    e.set_option("keep_preclean", 1)
    e.set_option("debug_stall_level", 9)
    e.set_option("debug_wait_rerun", 1 if rerun_fails else 0)
    e.set_global_map(mountain_small)
    e.init_graph([15.0, 15.0, 0.0])
    st = e.stats()
    assert st["used_device_bfs"] == 1 and st["bfs_fallbacks"] == 0, (st, e.fallback_reason)
    assert st["bfs_ticket_reruns"] == 1, st["bfs_ticket_reruns"]
    assert st["bfs_host_levels"] == natural + (1 if rerun_fails else 0), (st["bfs_host_levels"], natural)
    o = oa.Oracle(**prm)
    o.set_sampler(21, 0, 16)
    o.set_global_map(mountain_small)
    assert o.init_graph([15.0, 15.0, 0.0])
    assert_graph_equal(e.graph("preclean"), o.graph(1), WEIGHT_TOL)
    assert_graph_equal(e.graph("global"), o.graph(0), WEIGHT_TOL)
    c = o.counters()
    assert st["trials"] == c["trials"] and st["samples"] == c["samples"]
    assert st["created_nodes"] == c["created"] and st["invalid_nodes"] == c["invalid_created"]


@pytest.mark.parametrize("rerun_fails", [False, True], ids=["ticketed-repeat", "host-fallback"])
def test_failed_commit_lookback_is_repeated(oa, mountain_small, rerun_fails):
    """The commit's look-back scan waits for lower workgroups with a bound; when it runs out (another
    process's kernels on the card) the level's numbering is void: the grid reservations are taken back
    from the slots' outcomes and the launch is repeated with start tickets.  Only if the repeat fails as
    well (debug_wait_rerun) the whole build is redone by the host replay.  The hook makes one workgroup
    of level 7 give up."""
    prm = dict(oa.MOUNTAIN, sample_num=10)
    e = _engine(prm)
    e.set_sampler(21, 16)
    e.set_option("debug_lookback_level", 7)
    e.set_option("debug_wait_rerun", 1 if rerun_fails else 0)
    e.set_global_map(mountain_small)
    e.init_graph([15.0, 15.0, 0.0])
    st = e.stats()
    if rerun_fails:
        assert st["bfs_fallbacks"] == 1 and "look-back" in e.fallback_reason, (st, e.fallback_reason)
    else:
        assert st["bfs_fallbacks"] == 0 and st["used_device_bfs"] == 1, (st, e.fallback_reason)
        assert st["bfs_ticket_reruns"] == 1
    o = oa.Oracle(**prm)
    o.set_sampler(21, 0, 16)
    o.set_global_map(mountain_small)
    assert o.init_graph([15.0, 15.0, 0.0])
    assert_graph_equal(e.graph("global"), o.graph(0), WEIGHT_TOL)


@pytest.mark.parametrize("mode", ["presample_on", "tickets_always"])
def test_resolve_launch_variants_build_the_same_graph(oa, mountain_small, mode):
    """Two variants of the resolve launch must not change the graph: p_role workgroups (the pure sampling
    of the next level's nodes inside the launch that numbers them; off by default, it measured slower) and
    start tickets as workgroup indices for every launch (by default only the repeat of a launch whose wait
    ran out is ticketed)."""
    prm = dict(MOUNTAIN_S16)
    e = _engine(prm)
    e.set_sampler(7, 16)
    e.set_global_map(mountain_small)
    e.init_graph([15.0, 15.0, 0.0])
    g0 = e.graph("global")
    st0 = e.stats()
    assert st0["used_device_bfs"] == 1 and st0["presampled_nodes"] == 0
    e.close()
    e = _engine(prm)
    e.set_sampler(7, 16)
    e.set_option("presample" if mode == "presample_on" else "resolve_tickets", 1)
    e.set_global_map(mountain_small)
    e.init_graph([15.0, 15.0, 0.0])
    g1 = e.graph("global")
    st1 = e.stats()
    assert st1["used_device_bfs"] == 1 and st1["bfs_fallbacks"] == 0
    if mode == "presample_on":  # the samples of (nearly) every node really came from a p_role workgroup
        assert st1["presampled_nodes"] > 0.8 * st1["expanded_nodes"], (st1["presampled_nodes"], st1["expanded_nodes"])
    assert_graph_equal(g1, g0, 0.0)
    for k in ("trials", "samples", "created_nodes", "invalid_nodes", "bytes_sample_kernel"):
        assert st0[k] == st1[k], (k, st0[k], st1[k])


def test_statistics_equal_across_repeated_builds(mountain_small):
    """The expansion statistics are summed on the device after the level loop; the host must read them
    only when that kernel has finished (three builds on one engine give the same numbers)."""
    e = _engine(dict(MOUNTAIN_S16))
    e.set_sampler(7, 16)
    keys = ("trials", "samples", "created_nodes", "invalid_nodes", "edge_evals_gpu",
            "bytes_sample_kernel", "bytes_spec_kernel", "bytes_spec_created", "bytes_edge_kernel")
    seen = []
    for _ in range(3):
        e.set_global_map(mountain_small)
        e.init_graph([15.0, 15.0, 0.0])
        st = e.stats()
        seen.append(tuple(st[k] for k in keys))
    assert seen[0] == seen[1] == seen[2], seen
    assert seen[0][0] > 0 and seen[0][5] > 0


def test_device_build_is_deterministic(oa, mountain_gentle):
    """k_level_resolve lets thousands of lanes decide concurrently; the outcome must not depend on
    timing.  Crowded configuration (S=24), five builds, all equal to the oracle."""
    prm = dict(oa.MOUNTAIN, sample_num=24)
    o = oa.Oracle(**prm)
    o.set_sampler(13, 0, 16)
    o.set_global_map(mountain_gentle)
    assert o.init_graph([15.0, 15.0, 0.0])
    go = o.graph(0)
    e = _engine(prm)
    e.set_sampler(13, 16)
    e.set_global_map(mountain_gentle)
    for _ in range(5):
        e.init_graph([15.0, 15.0, 0.0])
        assert e.stats()["used_device_bfs"] == 1, e.fallback_reason
        # the container history differs between the first and later builds (bucket counts persist),
        # so compare through the creation ids, which are container-independent
        ge = e.graph("global")
        assert ge.V == go.V and ge.E == go.E
        assert np.array_equal(np.sort(ge.cid), np.sort(go.cid))
        order_e, order_o = np.argsort(ge.cid), np.argsort(go.cid)
        assert np.array_equal(ge.xyz[order_e], go.xyz[order_o])
        assert np.array_equal(np.diff(ge.rowptr)[order_e], np.diff(go.rowptr)[order_o])


def test_sparse_call_log_builds_the_same_graph(oa, mountain_small):
    """Builds with expandGraph's step 3 keep LEVEL_STEP3_STRIDE call-log entries per sample slot (the slot's
    own call, then room for the neighbour calls of the node it creates).  The same layout forced onto a
    step-3-off configuration must change nothing: graph, statistics, oracle parity."""
    prm = dict(MOUNTAIN_S16)
    graphs = []
    for stride in (0, 1):
        e = _engine(prm)
        e.set_sampler(7, 16)
        e.set_option("keep_preclean", 1)
        e.set_option("debug_call_stride", stride)
        e.set_global_map(mountain_small)
        e.init_graph([15.0, 15.0, 0.0])
        st = e.stats()
        assert st["used_device_bfs"] == 1 and st["bfs_fallbacks"] == 0, e.fallback_reason
        graphs.append((e.graph("preclean"), e.graph("global"), st))
    assert_graph_equal(graphs[1][0], graphs[0][0], 0.0)
    assert_graph_equal(graphs[1][1], graphs[0][1], 0.0)
    for k in ("expanded_nodes", "trials", "samples", "created_nodes", "invalid_nodes", "edge_evals_gpu",
              "bytes_sample_kernel", "bytes_spec_kernel", "bytes_edge_kernel"):
        assert graphs[0][2][k] == graphs[1][2][k], k


@pytest.mark.parametrize("S", [48, 64])
def test_large_sample_num_stays_on_the_device(oa, mountain_gentle, S):
    """sample_num up to 64 (the level kernels' limit; trg.cpp:387 has none): a slot can have more earlier
    candidates in reach than its 16-lane row holds at once (64) -- the row then takes them in passes (the
    hash probed again per pass, the nearest created one carried on) instead of sending the build to the
    host replay.  Oracle parity, no fallback, repeated builds identical."""
    prm = dict(oa.MOUNTAIN, sample_num=S)
    o = oa.Oracle(**prm)
    o.set_sampler(13, 0, 16)
    o.set_global_map(mountain_gentle)
    assert o.init_graph([15.0, 15.0, 0.0])
    e = _engine(prm)
    e.set_sampler(13, 16)
    e.set_option("keep_preclean", 1)
    e.set_global_map(mountain_gentle)
    e.init_graph([15.0, 15.0, 0.0])
    st = e.stats()
    assert st["used_device_bfs"] == 1 and st["bfs_fallbacks"] == 0, (st["bfs_fallbacks"], e.fallback_reason)
    if S == 64:
        assert st["bfs_multipass_rows"] > 0  # the several-passes path really ran
    assert_graph_equal(e.graph("preclean"), o.graph(1), WEIGHT_TOL)
    assert_graph_equal(e.graph("global"), o.graph(0), WEIGHT_TOL)
    c = o.counters()
    assert st["trials"] == c["trials"] and st["samples"] == c["samples"]
    assert st["created_nodes"] == c["created"] and st["invalid_nodes"] == c["invalid_created"]
    g0 = e.graph("preclean")
    for _ in range(2):
        e.init_graph([15.0, 15.0, 0.0])
        assert e.stats()["bfs_fallbacks"] == 0
        assert_graph_equal(e.graph("preclean"), g0, 0.0)


@pytest.mark.parametrize("replay", ["device", "host"])
def test_uncertain_slope_gates_are_decided_by_host_libm(oa, mountain_small, replay):
    """The device calls the slope gate (trg.cpp:269-274) with an exact rational test and leaves a
    thin band to the host's libm atan2f.  Widening the band (test hook) sends many gates through
    that path -- including ones the device provisionally treated as open but that are gated, which
    makes the device path take the level's commit back and redo it.  Results must not change."""
    prm = dict(oa.MOUNTAIN)
    e = _engine(prm)
    e.set_sampler(7, 16)
    e.set_option("keep_preclean", 1)
    e.set_option("replay", replay)
    e.set_option("debug_gate_margin", 0.15)
    e.set_global_map(mountain_small)
    e.init_graph([15.0, 15.0, 0.0])
    st = e.stats()
    assert st["gate_uncertain"] > 20, st["gate_uncertain"]
    assert st["used_device_bfs"] == (1 if replay == "device" else 0), e.fallback_reason
    o = oa.Oracle(**prm)
    o.set_sampler(7, 0, 16)
    o.set_global_map(mountain_small)
    assert o.init_graph([15.0, 15.0, 0.0])
    assert o.counters()["wire_gate"] > 0
    assert_graph_equal(e.graph("preclean"), o.graph(1), WEIGHT_TOL)
    assert_graph_equal(e.graph("global"), o.graph(0), WEIGHT_TOL)


def test_speculative_sampling_bound_and_top_up(oa, mountain_small):
    """The next BFS level is sampled before the host knows its size: the launch is an upper bound
    and a follow-up launch covers the rest when the frontier outgrew it.  With the bound capped at
    3 nodes (test hook) nearly every level needs the follow-up; the graph must not change."""
    prm = dict(oa.MOUNTAIN, sample_num=9)
    e, o = _build_both(oa, prm, mountain_small, [15.0, 15.0, 0.0], seed=5)
    ref = e.graph("global")
    e2 = _engine(prm)
    e2.set_sampler(5, 16)
    e2.set_option("debug_spec_bound", 3)
    e2.set_global_map(mountain_small)
    e2.init_graph([15.0, 15.0, 0.0])
    assert e2.stats()["used_device_bfs"] == 1, e2.fallback_reason
    assert_graph_equal(e2.graph("global"), o.graph(0), WEIGHT_TOL)
    assert_graph_equal(e2.graph("global"), ref, 0.0)
    for k in ("trials", "samples", "created_nodes", "invalid_nodes"):
        assert e2.stats()[k] == e.stats()[k], k


def test_device_bfs_declining_falls_back_to_the_host_replay(oa, mountain_small):
    """Capacity overflows make the device-resident BFS decline; the engine then redoes the build
    with the sequential host replay.  Forced here at level 6 (test hook)."""
    prm = dict(oa.MOUNTAIN)
    e = _engine(prm)
    e.set_sampler(7, 16)
    e.set_option("debug_fallback_level", 6)
    e.set_global_map(mountain_small)
    e.init_graph([15.0, 15.0, 0.0])
    st = e.stats()
    assert st["used_device_bfs"] == 0 and st["bfs_fallbacks"] == 1, st
    assert "declined on request" in e.fallback_reason
    o = oa.Oracle(**prm)
    o.set_sampler(7, 0, 16)
    o.set_global_map(mountain_small)
    assert o.init_graph([15.0, 15.0, 0.0])
    assert_graph_equal(e.graph("global"), o.graph(0), WEIGHT_TOL)
