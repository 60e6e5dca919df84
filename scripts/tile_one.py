import sys, time, numpy as np
sys.path.insert(0,'trg-planner_amd')
import trg_planner
from trg_planner import synth, tiled
rank = int(sys.argv[1]); use_tile = int(sys.argv[2])
nx, ny, S = 3200, 3125, 16
core = tiled.tile_cores(2, 1, nx, ny)[rank]
win = tiled.tile_lattice_window(rank, 2, 1, nx, ny, 11)
cloud = synth.mountain_tile(*win, seed=20250418)
start = [0.5*float(core[0]+core[2]), 0.5*float(core[1]+core[3]), 0.0]
prm = dict(expand_dist=0.6, robot_size=0.3, sample_num=S, height_threshold=0.16, collision_threshold=0.1, update_collision_threshold=0.5, safety_factor=3.0, goal_tolerance=0.8)
e = trg_planner.Engine(**prm); e.set_sampler(7,16)
if use_tile: e.set_tile(core, epoch=rank)
time.sleep(max(0.0, 12.0 - (time.time() % 12.0)) if False else 0)
for it in range(4):
    e.set_global_map(cloud); t=time.time(); e.init_graph(start); dt=time.time()-t
    st=e.stats()
    print("rank",rank,"it",it, f"{dt*1e3:.1f} ms", "dev", st["used_device_bfs"], "spin", st["bfs_max_spin"], "levels", st["bfs_levels"], repr(e.fallback_reason), flush=True)
