"""Spatial-tile sharding of TRG construction across the GPUs of one node.

The reference builds one graph from one root with one FIFO (trg.cpp:372-454); that BFS has a global
order and does not shard without changing results.  What shards naturally
is space: every rank owns one terrain tile and builds the complete graph of that tile (own map
index, own root at the tile centre).  The tile-boundary edges are then stitched with two
all-gather-v exchanges (tiled.py); besides those, ranks only meet at the
barrier and in the throughput reduction.
"""
from __future__ import annotations

import math


def tile_layout(world_size):
    """(cols, rows) of the tile grid: as square as possible, cols >= rows."""
    rows = int(math.floor(math.sqrt(world_size)))
    while world_size % rows:
        rows -= 1
    return world_size // rows, rows


def tile_of_rank(rank, world_size, nx, ny, spacing=0.1):
    """Origin (metres), centre and generator seed offset of this rank's tile."""
    cols, rows = tile_layout(world_size)
    cx, cy = rank % cols, rank // cols
    origin = (cx * nx * spacing, cy * ny * spacing)
    centre = (origin[0] + 0.5 * nx * spacing, origin[1] + 0.5 * ny * spacing)
    return {"origin": origin, "centre": centre, "seed_offset": rank, "grid": (cols, rows),
            "cell": (cx, cy)}


def reduce_throughput(items, seconds, dist=None, device=None):
    """Whole-job (sum of items, max of seconds) over all ranks; plain values when not distributed."""
    import os
    if dist is None or not dist.is_initialized() or (
            dist.get_world_size() == 1 and not os.environ.get("TRG_FORCE_COLLECTIVES")):
        return float(items), float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    it = torch.tensor([float(items)], dtype=torch.float64, device=device)
    dist.all_reduce(it, op=dist.ReduceOp.SUM)
    return float(it.item()), float(t.item())
