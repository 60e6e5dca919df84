"""Tiled CPU oracle (test infrastructure): the same tile rule as trg_planner/tiled.py, with the
reference restatement (oracle/) doing every tile's BFS and every cross-edge evaluation."""
import numpy as np


def stitch_oracle(tiled, prm, cols, rows, cores, oracles):
    """Steps 2-4 of the tile rule on the oracles' current tile graphs."""
    graphs = [o.graph(0) for o in oracles]
    all_idx = [tiled.boundary_nodes(g.xyz, cores[t], cols, rows, t, prm["expand_dist"])
               for t, g in enumerate(graphs)]
    all_xyz = [np.ascontiguousarray(g.xyz[i], np.float32) for g, i in zip(graphs, all_idx)]
    ids, w, d = [], [], []
    for t, o in enumerate(oracles):
        i_, w_, d_ = tiled.stitch_local(t, all_idx, all_xyz, prm["expand_dist"],
                                        lambda p1, p2, o=o: o.edge_risk(p1, p2))
        ids.append(i_)
        w.append(w_)
        d.append(d_)
    stitched = (np.concatenate(ids, 0), np.concatenate(w, 0), np.concatenate(d, 0))
    return graphs, stitched, tiled.assemble_global(graphs, stitched)


def build_tiled_oracle(oa, synth, tiled, prm, cols, rows, nx, ny, halo_pts, seed, sampler_seed,
                       terrain=None, return_oracles=False):
    cores = tiled.tile_cores(cols, rows, nx, ny)
    terrain = terrain or {}
    oracles = []
    for t, core in enumerate(cores):
        win = tiled.tile_lattice_window(t, cols, rows, nx, ny, halo_pts)
        cloud = synth.mountain_tile(*win, seed=seed, **terrain)
        o = oa.Oracle(**prm)
        o.set_sampler(sampler_seed, 0, 16)
        o.set_tile(core, epoch=t)
        o.set_global_map(cloud)
        start = [0.5 * (core[0] + core[2]), 0.5 * (core[1] + core[3]), 0.0]
        assert o.init_graph(start), f"tile {t}: no root"
        oracles.append(o)
    graphs, stitched, G = stitch_oracle(tiled, prm, cols, rows, cores, oracles)
    if return_oracles:
        return graphs, stitched, G, oracles
    return graphs, stitched, G
