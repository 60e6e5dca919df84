"""Tiled CPU oracle (test infrastructure): the same tile rule as trg_planner/tiled.py, with the
reference restatement (oracle/) doing every tile's BFS and every cross-edge evaluation."""
import numpy as np


def stitch_oracle(tiled, prm, cols, rows, cores, oracles):
    """Steps 2-4 of the tile rule on the oracles' current tile graphs."""
    graphs = [o.graph(0) for o in oracles]
    graphs, stitched, G = stitch_oracle(tiled, prm, cols, rows, cores, oracles)
    if return_oracles:
        return graphs, stitched, G, oracles
    return graphs, stitched, G
