#!/usr/bin/env python3
"""Run on the GPU box: cost of the native tile-boundary stitch at C3 tile size.  Two C3-sized tiles of
one terrain (2 x 1) are built one after the other on the one GPU, then every native step
(trg_engine_stitch_boundary / _cross / _assemble) is timed per tile; the two exchanges are
concatenations here (what RCCL adds on a real node is two small all-gathers per exchange)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "trg-planner_amd"))
import trg_planner  # noqa: E402
from trg_planner import synth, tiled  # noqa: E402

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3200, 3125)
cols, rows, halo = 2, 1, 11
prm = dict(expand_dist=0.6, robot_size=0.3, sample_num=16, height_threshold=0.16, collision_threshold=0.1,
           update_collision_threshold=0.5, safety_factor=3.0, goal_tolerance=0.8)
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
cores = tiled.tile_cores(cols, rows, nx, ny)
engines = []
for t, core in enumerate(cores):
    cloud = synth.mountain_tile(*tiled.tile_lattice_window(t, cols, rows, nx, ny, halo), seed=20250418)
    e = trg_planner.Engine(**prm)
    e.set_sampler(7, 16)
    e.set_tile(core, epoch=t)
    e.set_global_map(cloud)
    e.init_graph([0.5 * float(core[0] + core[2]), 0.5 * float(core[1] + core[3]), 0.0])
    engines.append(e)
    print(f"tile {t}: V'={e.graph_sizes('global')[0]} E'={e.graph_sizes('global')[1]}", flush=True)

res = {"tile_points": nx * ny, "reps": []}
for rep in range(5):
    t_b, t_c, t_a = [], [], []
    recs = []
    for t, e in enumerate(engines):
        r = torch.empty((1 << 14, 4), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = e.stitch_boundary(cores[t], cols, rows, t, r.data_ptr(), r.shape[0])
        t_b.append(1e3 * (time.perf_counter() - t0))
        recs.append(r[:nb])
    all_rec = torch.cat(recs, 0).contiguous()
    rec_off = np.concatenate([[0], np.cumsum([int(r.shape[0]) for r in recs])]).astype(np.int32)
    node_off = np.concatenate([[0], np.cumsum([e.graph_sizes("global")[0] for e in engines])]).astype(np.int32)
    parts = []
    for t, e in enumerate(engines):
        ed = torch.empty((1 << 15, 6), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nc = e.stitch_cross(t, cols * rows, all_rec.data_ptr(), rec_off, ed.data_ptr(), ed.shape[0])
        t_c.append(1e3 * (time.perf_counter() - t0))
        parts.append(ed[:nc])
    all_edges = torch.cat(parts, 0).contiguous()
    for t, e in enumerate(engines):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.stitch_assemble(t, cols * rows, node_off, all_edges.data_ptr(), int(all_edges.shape[0]))
        t_a.append(1e3 * (time.perf_counter() - t0))
    res["reps"].append({"ms_boundary": t_b, "ms_cross": t_c, "ms_assemble": t_a,
                        "boundary_records": int(rec_off[-1]), "cross_edges": int(all_edges.shape[0])})
last = res["reps"][-1]
res["ms_per_tile_last_rep"] = [last["ms_boundary"][t] + last["ms_cross"][t] + last["ms_assemble"][t]
                               for t in range(len(engines))]
res["stitched_sizes"] = [e.graph_sizes("stitched") for e in engines]
print(json.dumps(res, indent=1))
