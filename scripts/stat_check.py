#!/usr/bin/env python3
"""Print the engine's byte / hit statistics of one mid-size build (GPU box); TRG_ENGINE_LIB selects the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "trg-planner_amd"))
import torch  # noqa
import trg_planner
from trg_planner import synth
MOUNTAIN = dict(expand_dist=0.6, robot_size=0.3, height_threshold=0.16, collision_threshold=0.1,
                update_collision_threshold=0.5, safety_factor=3.0, goal_tolerance=0.8)
NX, NY = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 1000)
cloud = synth.mountain_tile(0, NX, 0, NY, seed=20250418)
e = trg_planner.Engine(**dict(MOUNTAIN, sample_num=16))
e.set_sampler(7, 16)
for rep in range(int(os.environ.get("REPS", "1"))):
  e.set_global_map(cloud)
  e.init_graph([NX * 0.05, NY * 0.05, 0.0])
  st = e.stats()
  print({k: st[k] for k in ("bytes_sample_kernel", "bytes_spec_kernel", "bytes_spec_created", "bytes_edge_kernel",
                         "trials", "samples", "created_nodes", "invalid_nodes", "edge_evals_gpu", "bfs_levels")})
