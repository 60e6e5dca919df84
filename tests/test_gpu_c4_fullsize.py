"""GPU: BASELINE config 4 at its real size -- the 10 M-point C3 cloud cut 2 x 2, every tile built with its
core restriction (one after the other on the single GPU of the test box: per-tile work and stitch rule are
exactly what every rank runs; the collective itself is covered by tests/test_dist_gloo.py), native stitch --
against the golden digest of the TILED CPU ORACLE's assembled global graph (tests/golden/c4_digest.json,
made once by scripts/fullscale_parity_c4.py on a GPU box: the oracle needs ~3 minutes at this size).

Bit-exact: V, E, tile offsets, rowptr, col, state, node xyz, edge dist (sha256 of the arrays), the cross-edge
count.  Weights: a fixed random sample of 65 536 edges within 1e-5 (clamp flips counted apart), zero-weight
count and weight sum of the whole graph."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sha(a, dt):
    return hashlib.sha256(np.ascontiguousarray(a.astype(dt, copy=False)).tobytes()).hexdigest()


def test_c4_fullsize_tiled_against_tiled_oracle_digest(synth):
    import trg_planner
    from trg_planner import tiled
    from conftest import weight_report
    dg = json.load(open(os.path.join(GOLD, "c4_digest.json")))
    ws = np.load(os.path.join(GOLD, "c4_w_sample.npz"))
    nx, ny, cols, rows = 3200, 3125, 2, 2
    prm = dict(expand_dist=0.6, robot_size=0.3, sample_num=16, height_threshold=0.16,
               collision_threshold=0.1, update_collision_threshold=0.5, safety_factor=3.0,
               goal_tolerance=0.8)
    cores, engines = [], []
    for t in range(cols * rows):
        core, win = tiled.split_tile(t, cols, rows, nx, ny, 11)
        cloud = synth.mountain_tile(*win, seed=20250418)
        e = trg_planner.Engine(**prm)
        e.set_sampler(7, 16)
        e.set_tile(core, epoch=t)
        e.set_global_map(cloud)
        del cloud
        e.init_graph([0.5 * float(core[0] + core[2]), 0.5 * float(core[1] + core[3]), 0.0])
        st = e.stats()
        assert st["used_device_bfs"] == 1 and st["bfs_fallbacks"] == 0, e.fallback_reason
        cores.append(core)
        engines.append(e)
    parts, cross = tiled.stitch_emulated(engines, cores, cols, rows)
    G = tiled.concat_stitched(parts)
    assert (G["V"], int(G["col"].size)) == (dg["V"], dg["E"]), (G["V"], G["col"].size)
    assert [int(x) for x in G["offsets"]] == dg["tile_offsets"]
    assert int(cross.shape[0]) == dg["cross_edges"]
    for name, arr, dt in (("rowptr", G["rowptr"], np.int64), ("col", G["col"], np.int64),
                          ("state", G["state"], np.int32), ("xyz", G["xyz"], np.float32),
                          ("dist", G["dist"], np.float32)):
        assert _sha(arr, dt) == dg["sha256"][name], name
    flips, others, mx = weight_report(G["w"][ws["idx"]], ws["w"], 1e-5)
    assert others == 0, (others, mx)
    known = int(dg.get("engine_at_digest_time", {}).get("clamp_flips") or 0)
    assert flips <= known, (flips, known)
    zero = int((G["w"] == 0).sum())
    assert abs(zero - dg["w_zero_edges"]) <= 2 * max(known, 1), (zero, dg["w_zero_edges"])
    wsum = float(G["w"].astype(np.float64).sum())
    assert abs(wsum - dg["w_sum"]) <= dg["E"] * 3e-6 + 0.2 * max(known, 1), (wsum, dg["w_sum"])
    # the tiled oracle's second witness (covariance accumulated in fp64, scripts/fullscale_parity_c4.py f64): every
    # weight of the assembled graph, the 5 431 stitched cross edges included, bit for bit
    wd = json.load(open(os.path.join(GOLD, "c4_witness_digest.json")))
    assert (G["V"], int(G["col"].size)) == (wd["V"], wd["E"])
    assert _sha(G["w"], np.float32) == wd["w_sha256_fp64_witness"]
    # the seam connects the tiles: every tile has edges into another one
    src = np.repeat(np.arange(G["V"]), np.diff(G["rowptr"]))
    tile_of = np.searchsorted(G["offsets"], np.arange(G["V"]), side="right") - 1
    xs = tile_of[src] != tile_of[G["col"]]
    assert set(tile_of[src][xs].tolist()) == set(range(cols * rows))
