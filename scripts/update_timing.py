#!/usr/bin/env python3
"""Run on the GPU box: where an updateGraph step spends its time at C3 size (TRG_TIMING=1 laps on stderr)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "trg-planner_amd"))
import torch  # noqa: F401,E402
import bench  # noqa: E402
import trg_planner  # noqa: E402
from trg_planner import synth  # noqa: E402

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3200, 3125)
cloud = synth.mountain_tile(0, nx, 0, ny, seed=20250418)
e = trg_planner.Engine(**dict(bench.MOUNTAIN, sample_num=16))
e.set_sampler(7, 16)
e.set_global_map(cloud)
e.init_graph([nx * 0.05, ny * 0.05, 0.0])
os.environ["TRG_TIMING"] = "1"
for k, (pose, obs) in enumerate(bench.obs_stream(6, (nx * 0.05 - 20.0, ny * 0.05 - 10.0), 20250418, (nx, ny))):
    t0 = time.perf_counter()
    e.set_local_map(pose, obs)
    t1 = time.perf_counter()
    e.update_graph()
    t2 = time.perf_counter()
    print(f"update {k}: set_local_map {1e3 * (t1 - t0):.2f} ms, update_graph {1e3 * (t2 - t1):.2f} ms", file=sys.stderr)
