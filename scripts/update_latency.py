#!/usr/bin/env python3
"""Latency of the incremental path (SURVEY section 8f row 1 / BASELINE config 5, second half):
a 1 M-point map, then a stream of local-map updates (20 m x 20 m crops of 40 k points around a pose
moving 0.5 m per step, with an injected obstacle) through setLocalMap + updateGraph.
Prints one JSON line; an evidence script, not part of bench.py's contract."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "trg-planner_amd"))
import trg_planner  # noqa: E402
from trg_planner import synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cloud = synth.mountain_tile(0, 1000, 0, 1000, seed=20250418)
prm = dict(expand_dist=0.6, robot_size=0.3, sample_num=16, height_threshold=0.16,
           collision_threshold=0.1, update_collision_threshold=0.5, safety_factor=3.0,
           goal_tolerance=0.8)
e = trg_planner.Engine(**prm)
e.set_sampler(7, 16)
e.set_global_map(cloud)
t0 = time.perf_counter()
e.init_graph([50.0, 50.0, 0.0])
t_init = time.perf_counter() - t0
g = e.graph("global")
lat_map, lat_upd, sizes = [], [], []
for k in range(steps):
    pose = (30.0 + 0.5 * k, 40.0 + 0.2 * k)
    m = (np.abs(cloud[:, 0] - pose[0]) < 10.0) & (np.abs(cloud[:, 1] - pose[1]) < 10.0)
    obs = cloud[m].copy()
    b = (np.abs(obs[:, 0] - pose[0] - 3.0) < 0.6) & (np.abs(obs[:, 1] - pose[1] - 1.0) < 0.6)
    obs[b, 2] += np.float32(1.0) * (np.arange(b.sum()) % 2).astype(np.float32)
    t0 = time.perf_counter()
    e.set_local_map(pose, obs)
    t1 = time.perf_counter()
    e.update_graph()
    t2 = time.perf_counter()
    lat_map.append(1e3 * (t1 - t0))
    lat_upd.append(1e3 * (t2 - t1))
    sizes.append(int(obs.shape[0]))
g2 = e.graph("global")
print(json.dumps({
    "map_points": int(cloud.shape[0]), "init_graph_ms": 1e3 * t_init, "V0": g.V, "E0": g.E,
    "updates": steps, "obs_points_mean": float(np.mean(sizes)),
    "set_local_map_ms_median": float(np.median(lat_map)), "update_graph_ms_median": float(np.median(lat_upd)),
    "update_graph_ms_p90": float(np.percentile(lat_upd, 90)), "V_end": g2.V, "E_end": g2.E}))
