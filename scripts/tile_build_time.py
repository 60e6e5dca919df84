import sys, time, numpy as np
sys.path.insert(0,'trg-planner_amd')
import trg_planner
from trg_planner import synth, tiled
nx, ny, S = 3200, 3125, 16
cols, rows = 2, 1
for rank in (0, 1):
    core = tiled.tile_cores(cols, rows, nx, ny)[rank]
    win = tiled.tile_lattice_window(rank, cols, rows, nx, ny, 11)
    cloud = synth.mountain_tile(*win, seed=20250418)
    start = [0.5*float(core[0]+core[2]), 0.5*float(core[1]+core[3]), 0.0]
    prm = dict(expand_dist=0.6, robot_size=0.3, sample_num=S, height_threshold=0.16, collision_threshold=0.1, update_collision_threshold=0.5, safety_factor=3.0, goal_tolerance=0.8)
    e = trg_planner.Engine(**prm); e.set_sampler(7,16); e.set_tile(core, epoch=rank)
    for it in range(2):
        e.set_global_map(cloud); t=time.time(); e.init_graph(start); dt=time.time()-t
        st=e.stats(); V,E=e.graph_sizes("global")
        print(rank, it, f"{dt*1e3:.1f} ms", V, E, "levels", st["bfs_levels"], "trials", st["trials"], "samples", st["samples"], "host_levels", st["bfs_host_levels"], "dev", st["used_device_bfs"], "loop", round(st["ms_bfs_loop"],1), "sample_ms", round(st["ms_sample_kernel"],1), e.fallback_reason, flush=True)
