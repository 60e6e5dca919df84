"""Tiled multi-GPU TRG build: one terrain tile per rank, boundary edges stitched over RCCL.

This is an EXTENSION of the reference algorithm, not a restatement: the reference grows one graph
from one root with one FIFO (trg.cpp:372-454), whose order is global and does not shard.  The
tiled build is defined as follows (the CPU oracle implements the same rule, tests/tiled_oracle.py):

1. Space is cut into a grid of core rectangles.  Rank t holds the map points of core_t plus a halo
   (>= 1.25*expand_dist + robot_size) and runs the reference BFS on them with
   one change: a sample outside core_t counts as a rejected draw, so every node of tile t lies in
   core_t.  Sampler: same seed, epoch = tile index.  cleanGraph runs per tile.
2. Boundary nodes = nodes closer than expand_dist to a core border shared with another tile.
   Every rank publishes them (all-gather-v: tile, local id, x, y, z).
3. For every pair (a in tile t, b in tile u, t < u) with ||a - b|| < expand_dist (fp32 norm as
   trg.cpp:414) rank t evaluates wireEdge's position-only part for (a, b) on its own map; the
   successful ones are published (second all-gather-v).
4. Every node appends its cross edges after its tile-local edges, ordered by the global id of the
   other endpoint.  Global id = tile offset (exclusive prefix of tile node counts) + local id.
"""
from __future__ import annotations

import numpy as np


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


def allgatherv(arr, dist=None, device=None):
    """All-gather of per-rank arrays with different leading sizes (RCCL has no native allgatherv:
    counts first, then padded payloads).  Returns the list of every rank's array."""
    arr = np.ascontiguousarray(arr)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [arr]
    import torch
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([arr.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = torch.cat(counts).cpu().tolist()  # one device round trip for all counts
    width = int(np.prod(arr.shape[1:])) if arr.ndim > 1 else 1
    pad = max(max(counts), 1)
    tdtype = torch.from_numpy(np.zeros(1, arr.dtype)).dtype
    buf = torch.zeros((pad, width), dtype=tdtype, device=dev)
    if arr.shape[0]:
        buf[:arr.shape[0]] = torch.from_numpy(arr.reshape(arr.shape[0], width)).to(dev)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    host = torch.stack(outs).cpu().numpy()  # one copy back for all ranks' payloads
    return [host[k, :c].reshape((c,) + arr.shape[1:]) for k, c in enumerate(counts)]


def cross_pairs(my_tile, records_per_tile, expand_dist):
    """Pairs (index into my boundary list, other tile, index into its boundary list) with my tile
    as the LOWER tile and fp32 planar distance < expand_dist, as three int arrays ordered by
    (other tile, my index, its index).  records: float32 (k, 3) xyz.
    A kd-tree prefilter (fp64, slightly enlarged radius) finds the candidates, the decision is the
    reference's fp32 expression (existing - sample).norm() < d, trg.cpp:414."""
    from scipy.spatial import cKDTree
    mine = records_per_tile[my_tile]
    d = np.float32(expand_dist)
    out_a, out_u, out_b = [], [], []
    if mine.shape[0]:
        lo, hi = mine[:, :2].min(0) - 1.01 * d, mine[:, :2].max(0) + 1.01 * d
        my_tree = None
        for u in range(my_tile + 1, len(records_per_tile)):
            other = records_per_tile[u]
            if other.shape[0] == 0:
                continue
            o_lo, o_hi = other[:, :2].min(0), other[:, :2].max(0)
            if (o_lo > hi).any() or (o_hi < lo).any():
                continue  # bounding boxes farther apart than expand_dist: no pair possible
            if my_tree is None:
                my_tree = cKDTree(mine[:, :2].astype(np.float64))
            pairs = my_tree.sparse_distance_matrix(cKDTree(other[:, :2].astype(np.float64)),
                                                   float(expand_dist) * 1.001 + 1e-6,
                                                   output_type="coo_matrix")
            ia, ib = pairs.row.astype(np.int64), pairs.col.astype(np.int64)
            if ia.size == 0:
                continue
            dx = mine[ia, 0] - other[ib, 0]
            dy = mine[ia, 1] - other[ib, 1]
            dist = np.sqrt(dx * dx + dy * dy, dtype=np.float32)  # fp32, no FMA
            keep = dist < d
            ia, ib = ia[keep], ib[keep]
            order = np.lexsort((ib, ia))
            out_a.append(ia[order])
            out_u.append(np.full(order.size, u, np.int64))
            out_b.append(ib[order])
    if not out_a:
        z = np.zeros(0, np.int64)
        return z, z, z
    return np.concatenate(out_a), np.concatenate(out_u), np.concatenate(out_b)


def assemble_global(tile_graphs, stitched):
    """Global CSR from the tile graphs (objects with V, rowptr, col, w, dist, xyz, state) and the
    stitched cross edges (array rows: tile_a, lid_a, tile_b, lid_b; parallel arrays w, dist)."""
    sizes = [g.V for g in tile_graphs]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    V = int(offs[-1])
    ids, w, dist = stitched
    extra = [[] for _ in range(V)]
    for k in range(ids.shape[0]):
        ga = int(offs[ids[k, 0]] + ids[k, 1])
        gb = int(offs[ids[k, 2]] + ids[k, 3])
        extra[ga].append((gb, float(w[k]), float(dist[k])))
        extra[gb].append((ga, float(w[k]), float(dist[k])))
    rowptr = [0]
    col, ww, dd = [], [], []
    xyz = np.concatenate([g.xyz for g in tile_graphs], 0) if V else np.zeros((0, 3), np.float32)
    state = np.concatenate([g.state for g in tile_graphs], 0) if V else np.zeros(0, np.int32)
    for t, g in enumerate(tile_graphs):
        for i in range(g.V):
            a, b = int(g.rowptr[i]), int(g.rowptr[i + 1])
            col.extend((g.col[a:b].astype(np.int64) + offs[t]).tolist())
            ww.extend(g.w[a:b].tolist())
            dd.extend(g.dist[a:b].tolist())
            for (gb, w_, d_) in sorted(extra[int(offs[t]) + i]):
                col.append(gb)
                ww.append(w_)
                dd.append(d_)
            rowptr.append(len(col))
    return dict(V=V, xyz=xyz, state=state, rowptr=np.array(rowptr, np.int64),
                col=np.array(col, np.int64), w=np.array(ww, np.float32),
                dist=np.array(dd, np.float32), offsets=offs)


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


def stitch(my_tile, graph, core, cols, rows, expand_dist, edge_risk, dist=None, device=None):
    """Steps 2-3 for one rank (rank == tile).  `edge_risk(p1, p2) -> (status, n_pts, w, dist)`
    evaluates wireEdge's position-only part on this rank's map (the engine's edge_risk_batch).
    Returns (ids[k,4] int32, w[k], dist[k]) of ALL ranks' stitched edges, identical on every rank,
    and the number of boundary records exchanged."""
    bidx = boundary_nodes(graph.xyz, core, cols, rows, my_tile, expand_dist)
    # exchange 1: boundary records (local id, x, y, z), 16 bytes each, one all-gather-v
    # (carried as int32 words: the payload is only ever copied, never computed on)
    rec = np.empty((bidx.shape[0], 4), np.int32)
    rec[:, 0] = bidx
    rec[:, 1:] = np.ascontiguousarray(graph.xyz[bidx], dtype=np.float32).view(np.int32)
    all_rec = allgatherv(rec, dist, device)
    if len(all_rec) == 1:  # single process: nothing to stitch against
        z = np.zeros(0, np.float32)
        return (np.zeros((0, 4), np.int32), z, z), 0
    all_idx = [np.ascontiguousarray(r[:, 0]) for r in all_rec]
    all_xyz = [np.ascontiguousarray(r[:, 1:]).view(np.float32) for r in all_rec]
    ids, w, d = stitch_local(my_tile, all_idx, all_xyz, expand_dist, edge_risk)
    # exchange 2: the cross edges this tile owns (tile_a, id_a, tile_b, id_b, weight, dist), 24 bytes
    out = np.empty((ids.shape[0], 6), np.int32)
    out[:, :4] = ids
    out[:, 4] = np.ascontiguousarray(w, dtype=np.float32).view(np.int32)
    out[:, 5] = np.ascontiguousarray(d, dtype=np.float32).view(np.int32)
    g = np.concatenate(allgatherv(out, dist, device), 0)
    return (np.ascontiguousarray(g[:, :4]), np.ascontiguousarray(g[:, 4]).view(np.float32),
            np.ascontiguousarray(g[:, 5]).view(np.float32)), int(sum(a.shape[0] for a in all_idx))
