#!/usr/bin/env python3
"""Edge-kernel microbenchmark: evaluates the edges of a real graph (1 M-point tile) again through
trg_engine_edge_risk_batch; run under `rocprofv3 --kernel-trace --stats` to read k_edges' time.
Used with the TRG_EDGE_STAGE_CUT profiling builds to attribute the kernel's cost to its stages."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "trg-planner_amd"))
import trg_planner  # noqa: E402
from trg_planner import synth  # noqa: E402

lib = os.environ.get("TRG_ENGINE_LIB")
cloud = synth.mountain_tile(0, 1000, 0, 1000, seed=20250418)
prm = dict(expand_dist=0.6, robot_size=0.3, sample_num=16, height_threshold=0.16,
           collision_threshold=0.1, update_collision_threshold=0.5, safety_factor=3.0,
           goal_tolerance=0.8)
pairs = os.path.join(ROOT, "gpurun_out", "edge_pairs.npz")
e = trg_planner.Engine(**prm)
e.set_sampler(7, 16)
e.set_global_map(cloud)
if os.path.exists(pairs):
    d = np.load(pairs)
    p1, p2 = d["p1"], d["p2"]
else:
    e.init_graph([50.0, 50.0, 0.0])
    g = e.graph("global")
    src = np.repeat(np.arange(g.V), np.diff(g.rowptr))
    p1, p2 = g.xyz[src], g.xyz[g.col]
    # plus as many node -> random nearby node pairs (most of them fail somewhere along the way)
    rng = np.random.default_rng(0)
    q = g.xyz[src].copy()
    ang = rng.uniform(0, 2 * np.pi, q.shape[0])
    q[:, 0] += 0.7 * np.cos(ang)
    q[:, 1] += 0.7 * np.sin(ang)
    q[:, 2] = e.nearest_z(q[:, :2])
    p1 = np.concatenate([p1, g.xyz[src]]).astype(np.float32)
    p2 = np.concatenate([p2, q]).astype(np.float32)
    os.makedirs(os.path.dirname(pairs), exist_ok=True)
    np.savez(pairs, p1=p1, p2=p2)
t0 = time.time()
for _ in range(3):
    st, npts, w, d = e.edge_risk(p1, p2)
print(f"{p1.shape[0]} edges x3 in {time.time() - t0:.3f}s; ok {(st == 0).mean():.3f}")
