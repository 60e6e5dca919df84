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
