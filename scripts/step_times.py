#!/usr/bin/env python3
"""Per-build wall times of consecutive C3 builds on one engine (GPU box): how long the warm-up lasts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "trg-planner_amd"))
import torch
import trg_planner
from trg_planner import synth, tiling
if os.environ.get('TRG_PIN_CPUS'):
    lo, hi = os.environ['TRG_PIN_CPUS'].split('-')
    os.sched_setaffinity(0, set(range(int(lo), int(hi) + 1)))
elif not os.environ.get('TRG_NO_PIN'):
    print('numa node', tiling.pin_to_gpu_numa(0), flush=True)
MOUNTAIN = dict(expand_dist=0.6, robot_size=0.3, height_threshold=0.16, collision_threshold=0.1,
                update_collision_threshold=0.5, safety_factor=3.0, goal_tolerance=0.8)
cloud = synth.mountain_tile(0, 3200, 0, 3125, seed=20250418)
d = torch.from_numpy(cloud).cuda()
e = trg_planner.Engine(**dict(MOUNTAIN, sample_num=16))
e.set_sampler(7, 16)
for k in range(16):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.set_global_map_device(d.data_ptr(), cloud.shape[0], 3)
    e.init_graph([160.0, 156.25, 0.0])
    dt = time.perf_counter() - t0
    st = e.stats()
    print(f"build {k}: {dt * 1e3:7.2f} ms  loop {st['ms_bfs_loop']:.2f} deferred {st['ms_deferred']:.2f} finalize {st['ms_finalize_host']:.2f} map {st['ms_set_map_total']:.2f} wait {st['ms_wait_gpu']:.2f}", flush=True)
