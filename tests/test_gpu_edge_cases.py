"""GPU: the error and edge behaviour of the reference's hot-path entry points, through the C ABI:
empty / tiny / degenerate maps, a start pose without a collision-free root, queries on an empty
graph, zero-length edges, and the JSON round trip of saveGraph / loadPrebuiltGraph."""
import numpy as np
import pytest

from conftest import assert_graph_equal

pytestmark = pytest.mark.gpu


def _engine(oa, **kw):
    import trg_planner
    e = trg_planner.Engine(**dict(oa.MOUNTAIN, **kw))
    e.set_sampler(7, 16)
    return e


def test_no_map_and_empty_graph_errors(oa):
    from trg_planner._engine import TrgError
    e = _engine(oa)
    with pytest.raises(TrgError) as ei:            # assert "Map is empty", trg.cpp:42
        e.init_graph([0.0, 0.0, 0.0])
    assert ei.value.status == 2
    with pytest.raises(TrgError) as ei:            # planning on an empty graph
        e.plan((0.0, 0.0), (1.0, 1.0, 0.0))
    assert ei.value.status == 5
    with pytest.raises(TrgError) as ei:            # elevation lookup without a map
        e.nearest_z(np.zeros((1, 2), np.float32))
    assert ei.value.status == 2
    # isCollision on an empty map: kd_nearest_range finds nothing -> collision (trg.cpp:749-752)
    flag, cnt, n = e.is_collision(np.zeros((3, 2), np.float32))
    assert flag.tolist() == [1, 1, 1] and n.tolist() == [0, 0, 0]
    g = e.graph("global")
    assert g.V == 0 and g.E == 0 and g.rowptr.tolist() == [0]


def test_start_without_a_collision_free_root(oa, synth):
    """initGraph tries start + (d, 0) and up to 100 random re-tries (trg.cpp:44-56); on a cloud
    where every disc collides the reference prints 'Failed to generate root node' and exits --
    here TRG_ERR_NO_ROOT; the oracle agrees that no graph can be built."""
    from trg_planner._engine import TrgError
    rng = np.random.default_rng(0)
    # a 'staircase' cloud: neighbouring points differ by far more than height_threshold
    ii, jj = np.meshgrid(np.arange(60), np.arange(60), indexing="ij")
    z = ((ii + jj) % 2).astype(np.float32) * 2.0
    cloud = np.stack([ii.ravel() * 0.1, jj.ravel() * 0.1, z.ravel()], 1).astype(np.float32)
    cloud = cloud[rng.permutation(cloud.shape[0])]
    e = _engine(oa)
    e.set_global_map(cloud)
    with pytest.raises(TrgError) as ei:
        e.init_graph([3.0, 3.0, 0.0])
    assert ei.value.status == 3
    o = oa.Oracle(**oa.MOUNTAIN)
    o.set_sampler(7, 0, 16)
    o.set_global_map(cloud)
    assert not o.init_graph([3.0, 3.0, 0.0])
    assert e.graph("global").V == 0


@pytest.mark.parametrize("replay", ["device", "host"])
def test_tiny_and_degenerate_maps(oa, replay):
    """Maps of a handful of points (the whole graph is the root, or a few nodes) and a map with
    exact duplicates: same graph as the oracle."""
    rng = np.random.default_rng(2)
    flat = np.stack([rng.uniform(0, 2.5, 700), rng.uniform(0, 2.5, 700), np.zeros(700)], 1).astype(np.float32)
    dup = np.concatenate([flat, flat[:200]])                       # exact duplicates
    three = np.asarray([[1.0, 1.0, 0.0], [1.1, 1.0, 0.0], [1.0, 1.1, 0.0]], np.float32)
    for cloud, start in ((flat, [0.6, 1.2, 0.0]), (dup, [0.6, 1.2, 0.0]), (three, [0.4, 1.0, 0.0])):
        e = _engine(oa)
        e.set_option("replay", replay)
        e.set_option("keep_preclean", 1)
        e.set_global_map(cloud)
        o = oa.Oracle(**oa.MOUNTAIN)
        o.set_sampler(7, 0, 16)
        o.set_global_map(cloud)
        ok_o = o.init_graph(start)
        try:
            e.init_graph(start)
            ok_e = True
        except Exception:
            ok_e = False
        assert ok_e == bool(ok_o)
        if ok_e:
            assert_graph_equal(e.graph("preclean"), o.graph(1), 1e-5)
            assert_graph_equal(e.graph("global"), o.graph(0), 1e-5)


def test_zero_length_and_long_edges(oa, mountain_gentle):
    """wireEdge's position-only part on degenerate pairs: identical endpoints (the segment walk never
    runs, the gather is a circle), and pairs far longer than any edge the build tries (general path
    of the kernel: more than six walk discs, box larger than the LDS tile)."""
    e = _engine(oa)
    e.set_global_map(mountain_gentle)
    o = oa.Oracle(**oa.MOUNTAIN)
    o.set_global_map(mountain_gentle)
    rng = np.random.default_rng(5)
    p = np.concatenate([rng.uniform(5, 25, (40, 2)), np.zeros((40, 1))], 1).astype(np.float32)
    p[:, 2] = e.nearest_z(p[:, :2])
    q = p.copy()
    ang = rng.uniform(0, 2 * np.pi, 20)
    q[20:, 0] += (3.0 * np.cos(ang)).astype(np.float32)            # 3 m: ~20 walk discs
    q[20:, 1] += (3.0 * np.sin(ang)).astype(np.float32)
    q[20:, 2] = e.nearest_z(q[20:, :2])
    st, npts, w, d = e.edge_risk(p, q)
    so, no, wo, do = o.edge_risk(p, q)
    assert np.array_equal(st, so)
    ok = st == 0
    assert np.array_equal(npts[ok], no[ok])
    assert np.array_equal(d.view(np.uint32), do.view(np.uint32))
    assert float(np.abs(w - wo).max()) <= 1e-5


def test_json_round_trip(oa, mountain_gentle, tmp_path):
    """saveGraph / loadPrebuiltGraph (trg.cpp:66-177): the reloaded graph plans the same paths."""
    e = _engine(oa)
    e.set_global_map(mountain_gentle)
    e.init_graph([15.0, 15.0, 0.0])
    g0 = e.graph("global")
    path = tmp_path / "graph"                      # no extension: ".json" is appended (trg.cpp:135-137)
    e.save_json(path)
    assert (tmp_path / "graph.json").exists()
    e2 = _engine(oa)
    e2.set_global_map(mountain_gentle)
    e2.load_json(tmp_path / "graph.json")
    g1 = e2.graph("global")
    assert g1.V == g0.V and g1.E == g0.E
    # ids are preserved by the file; adjacency as sets (the file lists edges per node)
    src0 = np.repeat(np.arange(g0.V), np.diff(g0.rowptr))
    src1 = np.repeat(np.arange(g1.V), np.diff(g1.rowptr))
    assert set(zip(src0.tolist(), g0.col.tolist())) == set(zip(src1.tolist(), g1.col.tolist()))
    assert np.allclose(g0.xyz, g1.xyz, atol=1e-4)  # the reference writes %f-style decimals
    for s, g in (((6.0, 7.0), (24.0, 22.0, 0.0)), ((20.0, 5.0), (8.0, 25.0, 0.0))):
        p0, i0 = e.plan(s, g)
        p1, i1 = e2.plan(s, g)
        assert p0.shape == p1.shape and p0.shape[0] > 1
        assert abs(i0.path_length - i1.path_length) < 1e-3


@pytest.mark.parametrize("replay", ["device", "host"])
@pytest.mark.parametrize("spacing", [0.04, 0.25])
def test_dense_and_sparse_maps(oa, synth, replay, spacing):
    """Map densities far from the benchmark's 0.1 m lattice: at 0.04 m a disc holds ~180 points and
    an edge's box far more than the LDS tiles take (global-memory fallbacks of the disc query, the
    median selection and the edge gather); at 0.25 m discs hold a handful of points and many edges
    have fewer than three ellipse points."""
    n = int(12.0 / spacing)
    cloud = synth.mountain_cloud(n, n, seed=8, spacing=spacing, jitter=0.2 * spacing, amplitude=2.0,
                                 wavelength=15.0)
    prm = dict(oa.MOUNTAIN, sample_num=9)
    start = [6.0, 6.0, 0.0]
    e = _engine(oa, sample_num=9)
    e.set_option("replay", replay)
    e.set_option("keep_preclean", 1)
    e.set_global_map(cloud)
    o = oa.Oracle(**prm)
    o.set_sampler(7, 0, 16)
    o.set_global_map(cloud)
    ok_o = bool(o.init_graph(start))
    try:
        e.init_graph(start)
        ok_e = True
    except Exception:
        ok_e = False
    assert ok_e == ok_o
    if ok_e:
        go = o.graph(1)
        assert_graph_equal(e.graph("preclean"), go, 1e-5)
        assert_graph_equal(e.graph("global"), o.graph(0), 1e-5)
        rng = np.random.default_rng(3)
        q = rng.uniform(1, 11, (400, 2)).astype(np.float32)
        fe, ce, ne = e.is_collision(q)
        fo, co, no = o.is_collision(q)
        assert np.array_equal(fe, fo) and np.array_equal(ce, co) and np.array_equal(ne, no)


def test_engine_reuse_across_maps_of_different_size(oa, synth):
    """One engine, three maps in a row (small -> large -> small, different extents and sample
    counts are not possible on one engine, so only the maps change): every device buffer is sized on
    demand and reused; each build must equal a fresh engine's build of the same map (compared through
    the creation ids, because cleanGraph's renumbering depends on the container's bucket history)."""
    import trg_planner
    maps = [synth.mountain_cloud(200, 200, seed=1), synth.mountain_cloud(500, 450, seed=2, origin=(-7.0, 3.0)),
            synth.mountain_cloud(150, 260, seed=3, origin=(40.0, -12.0))]
    starts = [[10.0, 10.0, 0.0], [18.0, 25.0, 0.0], [47.5, 1.0, 0.0]]
    prm = dict(oa.MOUNTAIN, sample_num=12)
    shared = trg_planner.Engine(**prm)
    shared.set_sampler(9, 16)
    for cloud, start in zip(maps, starts):
        shared.set_global_map(cloud)
        shared.init_graph(start)
        assert shared.stats()["used_device_bfs"] == 1, shared.fallback_reason
        fresh = trg_planner.Engine(**prm)
        fresh.set_sampler(9, 16)
        fresh.set_global_map(cloud)
        fresh.init_graph(start)
        a, b = shared.graph("global"), fresh.graph("global")
        assert a.V == b.V and a.E == b.E and a.V > 100
        oa_, ob_ = np.argsort(a.cid), np.argsort(b.cid)
        assert np.array_equal(a.cid[oa_], b.cid[ob_])
        assert np.array_equal(a.xyz[oa_].view(np.uint32), b.xyz[ob_].view(np.uint32))
        assert np.array_equal(np.diff(a.rowptr)[oa_], np.diff(b.rowptr)[ob_])
        # same edges (as creation-id pairs) with the same weights
        def edges(g):
            src = np.repeat(np.arange(g.V), np.diff(g.rowptr))
            key = g.cid[src].astype(np.int64) * (1 << 32) + g.cid[g.col]
            o = np.argsort(key)
            return key[o], g.w[o], g.dist[o]
        ka, wa, da = edges(a)
        kb, wb, db = edges(b)
        assert np.array_equal(ka, kb) and np.array_equal(wa.view(np.uint32), wb.view(np.uint32))
        assert np.array_equal(da.view(np.uint32), db.view(np.uint32))
