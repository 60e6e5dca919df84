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


def _cpu_busy_jiffies():
    busy = {}
    for ln in open("/proc/stat"):
        if ln.startswith("cpu") and ln[3].isdigit():
            f = ln.split()
            v = [int(x) for x in f[1:9]]
            busy[int(f[0][3:])] = v[0] + v[1] + v[2] + v[5] + v[6] + v[7]  # user nice system irq softirq steal
    return busy


def pin_to_gpu_numa(gpu_index, slot=0, nslots=1, width=8, probe_s=0.08):
    """Restrict the calling thread to ONE block of `width` neighbouring CPUs (a CCD) of the NUMA node
    the GPU hangs off: the block that was idlest over a short probe, among the blocks number
    slot, slot + nslots, ... (ranks of one node pass their local rank and count, so that they do not
    share a block).  The thread that drives a build launches ~15 small kernels per BFS level and
    polls a pinned counter block; when the scheduler is free to move it over 256 CPUs, or when it
    shares cores with another tenant's work, builds are up to 4 ms slower (launches take longer, the
    deferred pipeline falls behind); confined to an idle CCD every process runs like the best
    (measured at C3: 49.4-49.9 ms against 49.8-53.8).
    Returns (node, first cpu, last cpu), or None when the topology cannot be read."""
    import os
    import time
    try:
        import torch
        p = torch.cuda.get_device_properties(gpu_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        blocks = sorted({c // width for c in allowed})
        if not blocks:
            return None
        cand = blocks[slot % max(1, nslots)::max(1, nslots)] or blocks
        b0 = _cpu_busy_jiffies()
        time.sleep(probe_s)
        b1 = _cpu_busy_jiffies()
        load = {b: sum(b1.get(c, 0) - b0.get(c, 0) for c in allowed if c // width == b) for b in cand}
        b = min(cand, key=lambda k: (load[k], k))
        mine = {c for c in allowed if c // width == b}
        os.sched_setaffinity(0, mine)
        return node, min(mine), max(mine)
    except (OSError, ValueError, AttributeError, RuntimeError, IndexError):
        return None
