#!/usr/bin/env python3
"""BASELINE config C2 (1 M points, mountain.yaml, S = 7): the engine and the CPU oracle (one pinned core,
kd-tree = reference kdtree.c when oracle/_ref is present) on the same box; prints one JSON object."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "trg-planner_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa
import numpy as np
import trg_planner
from trg_planner import synth
import oracle_api as oa
cloud = synth.mountain_tile(0, 1000, 0, 1000, seed=20250418)
prm = dict(oa.MOUNTAIN, sample_num=7)
start = [50.0, 50.0, 0.0]
e = trg_planner.Engine(**prm)
e.set_sampler(7, 16)
e.set_global_map(cloud)
e.init_graph(start)
g_first = e.graph("global")  # (later builds on the same engine renumber differently: the reference's node
                             # container keeps its bucket count across builds, and so does the replica)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    e.set_global_map(cloud)
    e.init_graph(start)
    ts.append(time.perf_counter() - t0)
g = e.graph("global")
try:
    os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
except (AttributeError, OSError):
    pass
used_ref = oa.use_reference_kd(True)
tc = []
for _ in range(3):
    o = oa.Oracle(**prm)
    o.set_sampler(7, 0, 16)
    t0 = time.perf_counter()
    o.set_global_map(cloud)
    t1 = time.perf_counter()
    assert o.init_graph(start)
    t2 = time.perf_counter()
    go = o.graph(0)
    if not tc:
        same = bool(g_first.V == go.V and g_first.E == go.E and np.array_equal(g_first.col, go.col) and
                    np.array_equal(g_first.xyz.view(np.uint32), go.xyz.view(np.uint32)))
    tc.append((t2 - t0, t1 - t0, t2 - t1))
    o.close()
tc.sort()
print(json.dumps({"config": "C2: 1000x1000 pts, mountain.yaml, S=7", "V": int(g.V), "E": int(g.E),
                  "first_build_same_as_oracle": same,
                  "gpu_build_s_incl_pageable_upload_median": sorted(ts)[len(ts) // 2],
                  "cpu_build_s_median": tc[1][0], "cpu_index_s": tc[1][1], "cpu_init_graph_s": tc[1][2],
                  "kd": "reference kdtree.c" if used_ref else "oracle/okd.c"}))
