"""CPU, world_size 2 over gloo: the multi-rank path of bench.py (tile assignment, barrier,
sum-of-items / max-of-time reduction).  The per-rank graph build itself needs a GPU and is covered
by the -m gpu tests; here every rank stands in a fake item count."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "trg-planner_amd"))
    from trg_planner import synth, tiling
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = tiling.tile_of_rank(rank, world, 40, 30)
    cloud = synth.mountain_cloud(40, 30, seed=100 + t["seed_offset"], origin=t["origin"])
    dist.barrier()
    items, secs = tiling.reduce_throughput(1000 * (rank + 1), 0.5 + 0.25 * rank, dist)
    q.put((rank, t["origin"], float(cloud[:, 0].min()), float(cloud[:, 0].max()), items, secs))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tiles_and_reduction():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, o0, lo0, hi0, items0, secs0), (r1, o1, lo1, hi1, items1, secs1) = out
    assert o0 != o1 and hi0 <= lo1 + 0.1            # disjoint tiles side by side
    assert items0 == items1 == 3000.0               # whole-job aggregate on every rank
    assert secs0 == secs1 == 0.75                   # max over ranks


def test_tile_layouts():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "trg-planner_amd"))
    from trg_planner import tiling
    assert tiling.tile_layout(1) == (1, 1)
    assert tiling.tile_layout(2) == (2, 1)
    assert tiling.tile_layout(4) == (2, 2)
    assert tiling.tile_layout(8) == (4, 2)
    seen = {tiling.tile_of_rank(r, 8, 3200, 3125)["origin"] for r in range(8)}
    assert len(seen) == 8


def _stitch_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "trg-planner_amd"))
    from trg_planner import tiled
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # ragged all-gather: rank r contributes r+2 rows (and rank 1 an empty second array)
    a = np.arange((rank + 2) * 3, dtype=np.float32).reshape(rank + 2, 3) + 100 * rank
    cat, counts = tiled.allgatherv_t(torch.from_numpy(a), dist)
    bounds = np.concatenate([[0], np.cumsum(counts)])
    got = [cat.numpy()[bounds[k]:bounds[k + 1]] for k in range(world)]
    ecat, ecounts = tiled.allgatherv_t(torch.zeros((0 if rank == 1 else 2, 4), dtype=torch.int32), dist)
    empty = [np.zeros((c, 4), np.int32) for c in ecounts]
    assert ecat.shape == (sum(ecounts), 4)
    # a two-tile seam: nodes on a line at y = 1, tile 0 left of x = 5, tile 1 right of it
    class G:  # minimal tile graph
        pass
    g = G()
    xs = (np.arange(8, dtype=np.float32) * 0.25 + (3.2 if rank == 0 else 5.05))
    g.xyz = np.stack([xs, np.ones_like(xs), np.zeros_like(xs)], 1).astype(np.float32)
    core = np.array([0, 0, 5, 10], np.float32) if rank == 0 else np.array([5, 0, 10, 10], np.float32)

    def fake_edge_risk(p1, p2):  # every candidate pair succeeds; weight encodes the pair
        d = np.sqrt((p1[:, 0] - p2[:, 0]) ** 2 + (p1[:, 1] - p2[:, 1]) ** 2).astype(np.float32)
        return np.zeros(len(p1), np.int32), np.full(len(p1), 9, np.int32), 0.1 + 0 * d, d

    (ids, w, d), nrec = tiled.stitch_host(rank, g, core, 2, 1, 0.6, fake_edge_risk, dist)
    q.put((rank, [x.tolist() for x in got], [x.shape for x in empty], ids.tolist(), d.tolist(), nrec))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allgatherv_and_stitch():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stitch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = out
    assert r0[1] == r1[1] and len(r0[1][0]) == 2 and len(r0[1][1]) == 3   # ragged payloads intact
    assert r0[2] == [(2, 4), (0, 4)]
    assert r0[3] == r1[3] and len(r0[3]) > 0          # every rank ends with the same stitched list
    assert all(t[0] == 0 and t[2] == 1 for t in r0[3])  # owned by the lower tile
    assert all(dd < 0.6 for dd in r0[4])
    assert r0[5] == r1[5] and r0[5] > 0
