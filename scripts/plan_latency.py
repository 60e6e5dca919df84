#!/usr/bin/env python3
"""Run on the GPU box: latency of planSafePath (host A* straight on the CSR of the GPU-built graph) after a
C3 build -- first query (it builds the host node grid), steady state, a batch of 5 -- with the oracle's
planSafePath on its own C3 graph timed beside it on the box's host cores, for 5 start/goal pairs in the
style of the reference's run_trg_planner.py:26-43 (its mountain pairs, scaled to the 320 m terrain).

usage: python scripts/plan_latency.py [--no-oracle] [nx ny]   -> gpurun_out/plan_latency.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "trg-planner_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import trg_planner  # noqa: E402
from trg_planner import synth  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
with_oracle = "--no-oracle" not in sys.argv
nx, ny = (int(args[0]), int(args[1])) if len(args) >= 2 else (3200, 3125)
S = 16
prm = dict(expand_dist=0.6, robot_size=0.3, sample_num=S, height_threshold=0.16, collision_threshold=0.1,
           update_collision_threshold=0.1, safety_factor=3.0, goal_tolerance=0.8)
cloud = synth.mountain_tile(0, nx, 0, ny, seed=20250418)
cx, cy = nx * 0.05, ny * 0.05
start_pose = [cx, cy, 0.0]
# the reference's five mountain pairs (metres around its map origin), stretched 6x around the terrain centre
ref_s = np.array([[-7.22, -7.54], [-2.07, -2.21], [13.04, -1.99], [17.96, 17.69], [-6.56, 4.59]], np.float32)
ref_g = np.array([[-9.97, 3.56], [7.52, 1.44], [14.43, 6.87], [9.49, 16.60], [3.11, -6.68]], np.float32)
scale = 6.0 * min(nx, ny) / 3125.0
starts = (ref_s * scale + np.array([cx, cy], np.float32)).astype(np.float32)
goals = np.concatenate([ref_g * scale + np.array([cx, cy], np.float32), np.zeros((5, 1), np.float32)], 1).astype(np.float32)

e = trg_planner.Engine(**prm)
e.set_sampler(7, 16)
e.set_global_map(cloud)
t0 = time.perf_counter()
e.init_graph(start_pose)
t_build = time.perf_counter() - t0
V, E = e.graph_sizes("global")
t0 = time.perf_counter()
p_first, i_first = e.plan(starts[0], goals[0])
t_first = time.perf_counter() - t0
per_pair = []
paths = []
for s, g in zip(starts, goals):
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        p, info = e.plan(s, g)
        ts.append(time.perf_counter() - t0)
    paths.append(p)
    per_pair.append({"points": int(p.shape[0]), "path_length_m": float(info.path_length),
                     "direct_dist_m": float(info.direct_dist), "ms_median": 1e3 * float(np.median(ts)),
                     "ms_min": 1e3 * float(np.min(ts))})
tb = []
for _ in range(10):
    t0 = time.perf_counter()
    batch = e.plan_batch(starts, goals)
    tb.append(time.perf_counter() - t0)
for (pb, _), p in zip(batch, paths):
    assert np.array_equal(pb.view(np.uint32), p.view(np.uint32))
res = {"workload": f"C3-style {nx}x{ny} = {cloud.shape[0]} points, S={S}", "V": V, "E": E,
       "engine_build_ms": 1e3 * t_build,
       "engine_first_query_ms": 1e3 * t_first,
       "engine_first_query_note": "first planSafePath after the build: host node grid from the node arrays (O(V)), "
                                  "then the search; no edge pool, no node tree",
       "engine_steady_ms_per_query": per_pair,
       "engine_batch5_ms_median": 1e3 * float(np.median(tb)),
       "host_cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")}
if with_oracle:
    import oracle_api as oa  # noqa: E402  (the checker, timed beside the engine)
    oa.use_reference_kd(True)
    o = oa.Oracle(**prm)
    o.set_sampler(7, 0, 16)
    t0 = time.perf_counter()
    o.set_global_map(cloud)
    t_idx = time.perf_counter() - t0
    print(f"oracle index {t_idx:.1f}s", flush=True)
    t0 = time.perf_counter()
    assert o.init_graph(start_pose)
    t_init = time.perf_counter() - t0
    print(f"oracle initGraph {t_init:.1f}s", flush=True)
    orc = []
    same = True
    for k, (s, g) in enumerate(zip(starts, goals)):
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            po, io = o.plan(s, g)
            ts.append(time.perf_counter() - t0)
        same = same and po.shape == paths[k].shape and np.array_equal(po.view(np.uint32), paths[k].view(np.uint32))
        orc.append({"points": int(po.shape[0]), "ms_median": 1e3 * float(np.median(ts))})
    res.update({"oracle_index_s": t_idx, "oracle_init_graph_s": t_init, "oracle_ms_per_query": orc,
                "paths_bit_equal_to_oracle": bool(same),
                "oracle_note": "oracle = CPU restatement of trg.cpp:537-690 (hash maps keyed by node id, heap of "
                               "pointers, kd-tree lookups), one host core"})
out = os.path.join(ROOT, "gpurun_out", "plan_latency.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
