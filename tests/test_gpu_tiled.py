"""GPU: tiled (multi-GPU style) build against the tiled CPU oracle.  The ranks are emulated one
after the other on the single GPU of the test box -- the per-tile work and the stitch rule are
exactly what every rank runs; the collective itself is covered by tests/test_dist_gloo.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("layout", [(2, 1), (2, 2)])
def test_tiled_build_matches_tiled_oracle(oa, synth, layout):
    import trg_planner
    from trg_planner import tiled
    import tiled_oracle
    cols, rows = layout
    nx = ny = 130
    halo = 11  # 1.1 m at 0.1 m spacing
    prm = dict(oa.MOUNTAIN)
    terrain = dict(amplitude=2.0, wavelength=20.0)
    seed, sseed = 77, 9
    cores = tiled.tile_cores(cols, rows, nx, ny)
    engines, graphs = [], []
    for t, core in enumerate(cores):
        win = tiled.tile_lattice_window(t, cols, rows, nx, ny, halo)
        cloud = synth.mountain_tile(*win, seed=seed, **terrain)
        e = trg_planner.Engine(**prm)
        e.set_sampler(sseed, 16)
        e.set_tile(core, epoch=t)
        e.set_global_map(cloud)
        e.init_graph([0.5 * (core[0] + core[2]), 0.5 * (core[1] + core[3]), 0.0])
        assert e.stats()["used_device_bfs"] == 1, e.fallback_reason
        engines.append(e)
        graphs.append(e.graph("global"))
    # the native stitch (trg_engine_stitch_*: boundary, cross, assemble on the GPU), ranks emulated
    parts, cross = tiled.stitch_emulated(engines, cores, cols, rows)
    G = tiled.concat_stitched(parts)
    stitched = (cross[:, :4],)

    o_graphs, o_stitched, OG = tiled_oracle.build_tiled_oracle(
        oa, synth, tiled, prm, cols, rows, nx, ny, halo, seed, sseed, terrain)
    # every node inside its core; tiles equal the oracle's tiles
    for t, (g, og) in enumerate(zip(graphs, o_graphs)):
        c = cores[t]
        assert ((g.xyz[:, 0] >= c[0]) & (g.xyz[:, 0] < c[2]) & (g.xyz[:, 1] >= c[1]) &
                (g.xyz[:, 1] < c[3])).all()
        assert g.V == og.V and np.array_equal(g.col, og.col) and np.array_equal(g.xyz, og.xyz)
    assert stitched[0].shape[0] > 10                      # the seam really got edges
    assert np.array_equal(stitched[0], o_stitched[0])           # same cross edges in the same order
    assert np.array_equal(np.concatenate([p.cid for p in parts]), np.arange(G["V"]))  # rows = global ids
    assert G["V"] == OG["V"] and np.array_equal(G["rowptr"], OG["rowptr"])
    assert np.array_equal(G["col"], OG["col"]) and np.array_equal(G["state"], OG["state"])
    assert np.array_equal(G["xyz"].view(np.uint32), OG["xyz"].view(np.uint32))
    assert np.array_equal(G["dist"].view(np.uint32), OG["dist"].view(np.uint32))
    assert float(np.abs(G["w"] - OG["w"]).max()) <= TOL
    # the global graph is symmetric and the seam connects the tiles
    src = np.repeat(np.arange(G["V"]), np.diff(G["rowptr"]))
    fwd = set(zip(src.tolist(), G["col"].tolist()))
    assert all((b, a) in fwd for a, b in fwd)
    offs = G["offsets"]
    tile_of = np.searchsorted(offs, np.arange(G["V"]), side="right") - 1
    assert (tile_of[src] != tile_of[G["col"]]).any()


def _stitch_engines(tiled, prm, cols, rows, cores, engines):
    parts, cross = tiled.stitch_emulated(engines, cores, cols, rows)
    return parts, cross, tiled.concat_stitched(parts)


def _assert_global_equal(G, OG):
    assert G["V"] == OG["V"] and np.array_equal(G["rowptr"], OG["rowptr"])
    assert np.array_equal(G["col"], OG["col"]) and np.array_equal(G["state"], OG["state"])
    assert np.array_equal(G["xyz"].view(np.uint32), OG["xyz"].view(np.uint32))
    assert np.array_equal(G["dist"].view(np.uint32), OG["dist"].view(np.uint32))
    assert float(np.abs(G["w"] - OG["w"]).max()) <= TOL


def test_tiled_build_with_streaming_updates(oa, synth):
    """BASELINE config 5 in emulation: a 2 x 1 tiling (each tile with its core restriction and its own
    sampler epoch), then a stream of local-map updates -- setLocalMap + updateGraph (trg.cpp:195-231,
    456-489) -- on the tile that holds the pose, which moves along the seam; after every update the
    tile graph and the re-stitched global graph must equal the tiled CPU oracle put through the same
    steps."""
    import trg_planner
    from trg_planner import tiled
    import tiled_oracle
    cols, rows = 2, 1
    nx = ny = 130
    halo = 11
    prm = dict(oa.MOUNTAIN, update_collision_threshold=0.2)
    terrain = dict(amplitude=2.0, wavelength=20.0)
    seed, sseed = 77, 9
    cores = tiled.tile_cores(cols, rows, nx, ny)
    engines, clouds = [], []
    for t, core in enumerate(cores):
        win = tiled.tile_lattice_window(t, cols, rows, nx, ny, halo)
        cloud = synth.mountain_tile(*win, seed=seed, **terrain)
        e = trg_planner.Engine(**prm)
        e.set_sampler(sseed, 16)
        e.set_tile(core, epoch=t)
        e.set_global_map(cloud)
        e.init_graph([0.5 * (core[0] + core[2]), 0.5 * (core[1] + core[3]), 0.0])
        engines.append(e)
        clouds.append(cloud)
    _, _, OG, oracles = tiled_oracle.build_tiled_oracle(
        oa, synth, tiled, prm, cols, rows, nx, ny, halo, seed, sseed, terrain, return_oracles=True)
    _, _, G = _stitch_engines(tiled, prm, cols, rows, cores, engines)
    _assert_global_equal(G, OG)

    # the pose moves inside tile 0, 1.5 m from the seam (x = 13 m); obs = a 6 m x 6 m crop of the
    # tile's own cloud with a raised block (an obstacle the global map did not have)
    t = 0
    poses = [(11.5, 5.0), (11.5, 5.5), (11.5, 6.0)]
    changed = False
    for k, pose in enumerate(poses):
        cl = clouds[t]
        m = (np.abs(cl[:, 0] - pose[0]) < 3.0) & (np.abs(cl[:, 1] - pose[1]) < 3.0)
        obs = cl[m].copy()
        b = (np.abs(obs[:, 0] - (pose[0] + 1.0)) < 0.5) & (np.abs(obs[:, 1] - (pose[1] + 1.2)) < 0.5)
        obs[b, 2] += np.float32(1.0) * (np.arange(b.sum()) % 2).astype(np.float32)
        engines[t].set_local_map(pose, obs)
        oracles[t].set_local_map(pose, obs)
        before = engines[t].graph("global")
        engines[t].update_graph()
        oracles[t].update_graph()
        ge, go = engines[t].graph("global"), oracles[t].graph(0)
        assert ge.V == go.V and ge.E == go.E, (k, ge.V, go.V, ge.E, go.E)
        assert np.array_equal(ge.col, go.col) and np.array_equal(ge.state, go.state)
        assert np.array_equal(ge.xyz.view(np.uint32), go.xyz.view(np.uint32))
        assert float(np.abs(ge.w - go.w).max()) <= TOL
        c = cores[t]
        assert ((ge.xyz[:, 0] >= c[0]) & (ge.xyz[:, 0] < c[2]) & (ge.xyz[:, 1] >= c[1]) &
                (ge.xyz[:, 1] < c[3])).all()          # re-expansion respects the core
        changed |= (ge.V != before.V) or (ge.E != before.E) or not np.array_equal(ge.state, before.state)
        _, _, G = _stitch_engines(tiled, prm, cols, rows, cores, engines)
        _, _, OG = tiled_oracle.stitch_oracle(tiled, prm, cols, rows, cores, oracles)
        _assert_global_equal(G, OG)
    assert changed  # the updates really did something to the tile graph
