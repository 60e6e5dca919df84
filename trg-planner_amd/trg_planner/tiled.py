"""Tiled multi-GPU TRG build: one terrain tile per rank, boundary edges stitched over RCCL.

This is an EXTENSION of the reference algorithm, not a restatement: the reference grows one graph
from one root with one FIFO (trg.cpp:372-454), whose order is global and does not shard.  The
tiled build is defined as follows (the CPU oracle implements the same rule, tests/tiled_oracle.py):

1. Space is cut into a grid of core rectangles.  Rank t holds the map points of core_t plus a halo
   (>= 1.25*expand_dist + robot_size) and runs the reference BFS on them with
   one change: a sample outside core_t counts as a rejected draw, so every node of tile t lies in
   core_t.  Sampler: same seed, epoch = tile index.  cleanGraph runs per tile.
2. Boundary nodes = nodes closer than expand_dist to a core border shared with another tile.
   Every rank publishes them (all-gather-v: local id, x, y, z).
3. For every pair (a in tile t, b in tile u, t < u) with ||a - b|| < expand_dist (fp32 norm as
   trg.cpp:414) rank t evaluates wireEdge's position-only part for (a, b) on its own map; the
   successful ones are published (second all-gather-v).
4. Every node appends its cross edges after its tile-local edges, ordered by the global id of the
   other endpoint.  Global id = tile offset (exclusive prefix of tile node counts) + local id.

Product path: stitch_device (the engine's native trg_engine_stitch_* steps on the GPU, device tensors
through the two exchanges).  The numpy functions below state the same rule a second time for the
tiled CPU oracle and the CPU gloo test.
"""
from __future__ import annotations

import os
import sys

import numpy as np

try:  # the exchanges ride on torch.distributed; torch's ROCm libraries must load before the engine's
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None


def tile_cores(cols, rows, nx, ny, spacing=0.1):
    """Core rectangles [x0, y0, x1, y1) of a cols x rows grid of nx x ny lattice tiles (fp32)."""
    cores = []
    for r in range(rows):
        for c in range(cols):
            cores.append(np.array([c * nx * spacing, r * ny * spacing,
                                   (c + 1) * nx * spacing, (r + 1) * ny * spacing], np.float32))
    return cores


def tile_lattice_window(tile, cols, rows, nx, ny, halo_pts):
    """Global lattice index window [ix0, ix1) x [iy0, iy1) of a tile's core + halo, clipped to the
    whole terrain."""
    c, r = tile % cols, tile // cols
    ix0 = max(0, c * nx - halo_pts)
    ix1 = min(cols * nx, (c + 1) * nx + halo_pts)
    iy0 = max(0, r * ny - halo_pts)
    iy1 = min(rows * ny, (r + 1) * ny + halo_pts)
    return ix0, ix1, iy0, iy1


def split_tile(tile, cols, rows, gx, gy, halo_pts, spacing=0.1):
    """Strong scaling: ONE gx x gy lattice cut into cols x rows tiles, lattice columns / rows split as
    evenly as they go.  Returns (core rectangle [x0, y0, x1, y1) fp32, lattice window of core + halo)."""
    c, r = tile % cols, tile // cols
    cx0, cx1 = (c * gx) // cols, ((c + 1) * gx) // cols
    cy0, cy1 = (r * gy) // rows, ((r + 1) * gy) // rows
    core = np.array([cx0 * spacing, cy0 * spacing, cx1 * spacing, cy1 * spacing], np.float32)
    win = (max(0, cx0 - halo_pts), min(gx, cx1 + halo_pts), max(0, cy0 - halo_pts), min(gy, cy1 + halo_pts))
    return core, win


def boundary_nodes(xyz, core, cols, rows, tile, dist):
    """Indices of the nodes closer than `dist` to a core side that has a neighbouring tile."""
    c, r = tile % cols, tile // cols
    x, y = xyz[:, 0], xyz[:, 1]
    m = np.zeros(xyz.shape[0], bool)
    d = np.float32(dist)
    if c > 0:
        m |= (x - core[0]) < d
    if c < cols - 1:
        m |= (core[2] - x) < d
    if r > 0:
        m |= (y - core[1]) < d
    if r < rows - 1:
        m |= (core[3] - y) < d
    return np.nonzero(m)[0].astype(np.int32)


def allgatherv_t(t, dist=None):
    """All-gather of per-rank torch tensors with different leading sizes ON THE DEVICE THEY LIVE ON
    (device tensors over RCCL, host tensors over gloo): counts first, then padded payloads (RCCL has
    no native allgatherv).  Returns (concatenation in rank order, counts list)."""
    import torch
    if dist is None or not dist.is_initialized() or (
            dist.get_world_size() == 1 and not os.environ.get("TRG_FORCE_COLLECTIVES")):
        return t, [int(t.shape[0])]  # (TRG_FORCE_COLLECTIVES: tests run the exchange with one rank)
    world = dist.get_world_size()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = torch.cat(counts).cpu().tolist()  # the one host round trip of the exchange
    pad = max(max(counts), 1)
    buf = torch.zeros((pad,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    buf[:t.shape[0]] = t
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return torch.cat([o[:c] for o, c in zip(outs, counts)], 0), counts


def native_comm(eng, dist):
    """The engine's own RCCL communicator over the ranks of `dist` (rank == tile): one rank draws the
    unique id, torch.distributed carries the 128 bytes to the others (any backend), every rank joins.
    Returns True when the engine now holds a communicator of dist's size."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if getattr(eng, "_comm_ranks", 0) == world:
        return True
    ids = [eng.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    eng.comm_init(ids[0], world, rank)
    return True


def stitch_device(eng, my_tile, core, cols, rows, dist=None, gather_dev=None):
    """Steps 2-4 for one rank (rank == tile) with the engine's native stitch (include/trg_engine.h,
    trg_engine_stitch_*): boundary extraction, pair search, cross-edge evaluation and the assembly of
    this tile's rows of the global graph run on the GPU.  With one rank per GPU the two exchanges run
    inside the engine as well (trg_engine_stitch_exchange: RCCL all-gathers on the engine's own
    communicator -- the path a C++ consumer uses; TRG_NATIVE_EXCHANGE=0 keeps the exchanges in
    torch.distributed); with host tensors (`gather_dev` = CPU: more ranks than GPUs, gloo rehearsal) or
    without a process group they are torch.distributed all-gathers of padded tensors.
    Returns dict(n_boundary, n_cross, node_offsets, backend)."""
    import torch
    ntiles = cols * rows
    if (dist is not None and dist.is_initialized() and gather_dev is None and dist.get_world_size() == ntiles
            and (ntiles > 1 or os.environ.get("TRG_FORCE_COLLECTIVES"))
            and os.environ.get("TRG_NATIVE_EXCHANGE", "1") != "0" and dist.get_rank() == my_tile):
        # The engine's communicator is set up once; whether that worked is agreed on by all ranks (a rank that
        # could not open RCCL or join must not leave the others waiting in the first all-gather): if any of
        # them failed, every rank keeps the exchanges in torch.distributed (below) for this engine.
        ok = getattr(eng, "_native_ok", None)
        if ok is None:
            from ._engine import TrgError
            try:
                native_comm(eng, dist)
                mine = 1
            except (TrgError, OSError) as ex:
                print(f"[trg tiled] rank {dist.get_rank()}: native RCCL exchange unavailable ({ex}); "
                      "the exchanges stay in torch.distributed", file=sys.stderr, flush=True)
                mine = 0
            flag = torch.tensor([mine], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = eng._native_ok = bool(int(flag.item()))
        if ok:
            nb, nc = eng.stitch_exchange(core, cols, rows)
            return dict(n_boundary=nb, n_cross=nc, node_offsets=None, backend="rccl-native")
    dev = torch.device("cuda", torch.cuda.current_device())
    gdev = dev if gather_dev is None else gather_dev
    bufs = getattr(eng, "_stitch_bufs", None)  # device scratch kept with the engine, grown on demand
    if bufs is None:
        bufs = eng._stitch_bufs = {"rec": torch.empty((1 << 14, 4), dtype=torch.int32, device=dev),
                                   "edges": torch.empty((1 << 15, 6), dtype=torch.int32, device=dev)}
    from ._engine import TrgError
    while True:
        rec = bufs["rec"]
        try:
            nb = eng.stitch_boundary(core, cols, rows, my_tile, rec.data_ptr(), rec.shape[0])
            break
        except TrgError as ex:
            if ex.status != 8:  # TRG_ERR_CAPACITY
                raise
            need = eng.stitch_boundary(core, cols, rows, my_tile)
            bufs["rec"] = torch.empty((2 * need, 4), dtype=torch.int32, device=dev)
    # exchange 1: the boundary records, and riding behind them one row with this tile's node count
    V = eng.graph_sizes("global")[0]
    send = torch.cat([rec[:nb], torch.tensor([[V, 0, 0, 0]], dtype=torch.int32, device=dev)], 0)
    got, counts = allgatherv_t(send.to(gdev), dist)
    if len(counts) != ntiles:  # single process: nothing to stitch against
        node_off = np.array([0, V], np.int32)
        eng.stitch_assemble(0, 1, node_off, None, 0)
        return dict(n_boundary=0, n_cross=0, node_offsets=node_off, backend="none")
    ends = np.cumsum(counts)
    got = got.to(dev)
    node_off = np.concatenate([[0], np.cumsum(got[torch.as_tensor(ends - 1, device=dev), 0].cpu().numpy())]
                              ).astype(np.int32)
    all_rec = torch.cat([got[e - c:e - 1] for e, c in zip(ends, counts)], 0).contiguous()
    rec_off = np.concatenate([[0], np.cumsum([c - 1 for c in counts])]).astype(np.int32)
    torch.cuda.synchronize()
    while True:
        edges = bufs["edges"]
        try:
            nc = eng.stitch_cross(my_tile, ntiles, all_rec.data_ptr(), rec_off, edges.data_ptr(), edges.shape[0])
            break
        except TrgError as ex:
            if ex.status != 8:
                raise
            bufs["edges"] = torch.empty((2 * edges.shape[0], 6), dtype=torch.int32, device=dev)
    all_edges, ecounts = allgatherv_t(edges[:nc].to(gdev), dist)
    all_edges = all_edges.to(dev).contiguous()
    torch.cuda.synchronize()
    eng.stitch_assemble(my_tile, ntiles, node_off, all_edges.data_ptr(), int(all_edges.shape[0]))
    return dict(n_boundary=int(rec_off[-1]), n_cross=int(all_edges.shape[0]), node_offsets=node_off,
                backend=(dist.get_backend() if dist is not None and dist.is_initialized() else "none"))


def concat_stitched(parts):
    """Global CSR from the tiles' stitched rows (objects with rowptr, col, w, dist, xyz, state in tile
    order) -- a concatenation: the rows already carry global column ids."""
    rowptr = [np.zeros(1, np.int64)]
    base = 0
    for g in parts:
        rowptr.append(g.rowptr[1:].astype(np.int64) + base)
        base += int(g.rowptr[-1])
    sizes = [g.V for g in parts]
    return dict(V=int(sum(sizes)), xyz=np.concatenate([g.xyz for g in parts], 0),
                state=np.concatenate([g.state for g in parts], 0), rowptr=np.concatenate(rowptr),
                col=np.concatenate([g.col for g in parts]).astype(np.int64),
                w=np.concatenate([g.w for g in parts]), dist=np.concatenate([g.dist for g in parts]),
                offsets=np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64))


# ---- reference implementation of the stitch rule in numpy (tests/tiled_oracle.py drives the CPU
# oracle through it; the product path is stitch_device above) ------------------------------------------
def cross_pairs(my_tile, records_per_tile, expand_dist):
    """Pairs (index into my boundary list, other tile, index into its boundary list) with my tile
    as the LOWER tile and fp32 planar distance < expand_dist -- the reference's expression
    (existing - sample).norm() < d, trg.cpp:414 -- ordered by (other tile, my index, its index)."""
    mine = records_per_tile[my_tile]
    d = np.float32(expand_dist)
    out_a, out_u, out_b = [], [], []
    for u in range(my_tile + 1, len(records_per_tile)):
        other = records_per_tile[u]
        if mine.shape[0] == 0 or other.shape[0] == 0:
            continue
        dx = mine[:, None, 0] - other[None, :, 0]
        dy = mine[:, None, 1] - other[None, :, 1]
        ia, ib = np.nonzero(np.sqrt(dx * dx + dy * dy, dtype=np.float32) < d)  # fp32, no FMA; row-major
        out_a.append(ia.astype(np.int64))
        out_u.append(np.full(ia.size, u, np.int64))
        out_b.append(ib.astype(np.int64))
    if not out_a:
        z = np.zeros(0, np.int64)
        return z, z, z
    return np.concatenate(out_a), np.concatenate(out_u), np.concatenate(out_b)


def assemble_global(tile_graphs, stitched):
    """Global CSR from the tile graphs (objects with V, rowptr, col, w, dist, xyz, state) and the
    stitched cross edges (array rows: tile_a, lid_a, tile_b, lid_b; parallel arrays w, dist): every
    node's cross edges follow its tile-local edges, ordered by the other end's global id."""
    sizes = [g.V for g in tile_graphs]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    V = int(offs[-1])
    ids, w, dist = stitched
    xyz = np.concatenate([g.xyz for g in tile_graphs], 0) if V else np.zeros((0, 3), np.float32)
    state = np.concatenate([g.state for g in tile_graphs], 0) if V else np.zeros(0, np.int32)
    # local edges: (row, rank inside the row, col, w, dist)
    rows_l = np.concatenate([np.repeat(np.arange(g.V, dtype=np.int64), np.diff(g.rowptr)) + offs[t]
                             for t, g in enumerate(tile_graphs)]) if V else np.zeros(0, np.int64)
    col_l = np.concatenate([g.col.astype(np.int64) + offs[t] for t, g in enumerate(tile_graphs)])
    w_l = np.concatenate([g.w for g in tile_graphs])
    d_l = np.concatenate([g.dist for g in tile_graphs])
    ga = offs[ids[:, 0]] + ids[:, 1] if ids.shape[0] else np.zeros(0, np.int64)
    gb = offs[ids[:, 2]] + ids[:, 3] if ids.shape[0] else np.zeros(0, np.int64)
    rows_x = np.concatenate([ga, gb])
    col_x = np.concatenate([gb, ga])
    w_x = np.concatenate([w, w]).astype(np.float32)
    d_x = np.concatenate([dist, dist]).astype(np.float32)
    ox = np.lexsort((col_x, rows_x))  # a row's cross edges by the other end's global id
    rows_all = np.concatenate([rows_l, rows_x[ox]])
    kind = np.concatenate([np.zeros(rows_l.size, np.int8), np.ones(ox.size, np.int8)])
    order = np.lexsort((np.arange(rows_all.size), kind, rows_all))  # stable: local first, then cross
    deg = np.bincount(rows_all, minlength=V)
    return dict(V=V, xyz=xyz, state=state, rowptr=np.concatenate([[0], np.cumsum(deg)]).astype(np.int64),
                col=np.concatenate([col_l, col_x[ox]])[order],
                w=np.concatenate([w_l, w_x[ox]])[order].astype(np.float32),
                dist=np.concatenate([d_l, d_x[ox]])[order].astype(np.float32), offsets=offs)


def stitch_local(my_tile, all_idx, all_xyz, expand_dist, edge_risk):
    """Step 3 for one tile given every tile's boundary records: the cross edges this tile owns
    (it is the lower tile of the pair).  Returns (ids[k,4] int32, w[k], dist[k])."""
    ia, iu, ib = cross_pairs(my_tile, all_xyz, expand_dist)
    if ia.size == 0:
        return np.zeros((0, 4), np.int32), np.zeros(0, np.float32), np.zeros(0, np.float32)
    p1 = all_xyz[my_tile][ia]
    p2 = np.empty_like(p1)
    id_b = np.empty(ia.size, np.int32)
    for u in np.unique(iu):
        sel = iu == u
        p2[sel] = all_xyz[u][ib[sel]]
        id_b[sel] = all_idx[u][ib[sel]]
    st, _, w, d = edge_risk(p1, p2)
    ok = st == 0
    ids = np.stack([np.full(ia.size, my_tile, np.int32), all_idx[my_tile][ia].astype(np.int32),
                    iu.astype(np.int32), id_b], 1)[ok]
    return ids, w[ok].astype(np.float32), d[ok].astype(np.float32)




def stitch_host(my_tile, graph, core, cols, rows, expand_dist, edge_risk, dist=None):
    """The stitch rule with numpy on the host (steps 2-3; what tests/tiled_oracle.py and the CPU
    world_size-2 gloo test run; the product path is stitch_device): `edge_risk(p1, p2) -> (status,
    n_pts, w, dist)` evaluates wireEdge's position-only part on this rank's map.  Returns
    (ids[k,4] int32, w[k], dist[k]) of ALL ranks' cross edges, identical on every rank, and the number
    of boundary records exchanged."""
    import torch
    bidx = boundary_nodes(graph.xyz, core, cols, rows, my_tile, expand_dist)
    rec = np.empty((bidx.shape[0], 4), np.int32)  # (local id, x, y, z) as int32 words: only ever copied
    rec[:, 0] = bidx
    rec[:, 1:] = np.ascontiguousarray(graph.xyz[bidx], dtype=np.float32).view(np.int32)
    all_rec, counts = allgatherv_t(torch.from_numpy(rec), dist)
    if len(counts) == 1:  # single process: nothing to stitch against
        z = np.zeros(0, np.float32)
        return (np.zeros((0, 4), np.int32), z, z), 0
    bounds = np.concatenate([[0], np.cumsum(counts)])
    all_rec = all_rec.numpy()
    per = [all_rec[bounds[k]:bounds[k + 1]] for k in range(len(counts))]
    all_idx = [np.ascontiguousarray(r[:, 0]) for r in per]
    all_xyz = [np.ascontiguousarray(r[:, 1:]).view(np.float32) for r in per]
    ids, w, d = stitch_local(my_tile, all_idx, all_xyz, expand_dist, edge_risk)
    out = np.empty((ids.shape[0], 6), np.int32)  # (tile_a, id_a, tile_b, id_b, weight, dist), 24 bytes
    out[:, :4] = ids
    out[:, 4] = np.ascontiguousarray(w, dtype=np.float32).view(np.int32)
    out[:, 5] = np.ascontiguousarray(d, dtype=np.float32).view(np.int32)
    g, _ = allgatherv_t(torch.from_numpy(out), dist)
    g = g.numpy()
    return (np.ascontiguousarray(g[:, :4]), np.ascontiguousarray(g[:, 4]).view(np.float32),
            np.ascontiguousarray(g[:, 5]).view(np.float32)), int(bounds[-1])


def stitch_emulated(engines, cores, cols, rows):
    """All ranks of a tiling one after the other on ONE GPU (tests, rehearsals): the same three
    native steps as stitch_device, the two exchanges replaced by concatenations.  Returns the list of
    every tile's stitched graph (rows with global column ids) and the number of cross edges."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    ntiles = cols * rows
    recs = []
    for t, e in enumerate(engines):
        nb = e.stitch_boundary(cores[t], cols, rows, t)
        r = torch.empty((max(nb, 1), 4), dtype=torch.int32, device=dev)
        if nb:
            e.stitch_boundary(cores[t], cols, rows, t, r.data_ptr(), nb)
        recs.append(r[:nb])
    all_rec = torch.cat(recs, 0).contiguous()
    rec_off = np.concatenate([[0], np.cumsum([int(r.shape[0]) for r in recs])]).astype(np.int32)
    node_off = np.concatenate([[0], np.cumsum([e.graph_sizes("global")[0] for e in engines])]).astype(np.int32)
    torch.cuda.synchronize()
    parts = []
    for t, e in enumerate(engines):
        nc = e.stitch_cross(t, ntiles, all_rec.data_ptr(), rec_off)
        ed = torch.empty((max(nc, 1), 6), dtype=torch.int32, device=dev)
        if nc:
            nc = e.stitch_cross(t, ntiles, all_rec.data_ptr(), rec_off, ed.data_ptr(), nc)
        parts.append(ed[:nc])
    all_edges = torch.cat(parts, 0).contiguous()
    torch.cuda.synchronize()
    out = []
    for t, e in enumerate(engines):
        e.stitch_assemble(t, ntiles, node_off, all_edges.data_ptr(), int(all_edges.shape[0]))
        out.append(e.graph("stitched"))
    return out, all_edges.cpu().numpy()
