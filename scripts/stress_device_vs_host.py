import sys, numpy as np
sys.path.insert(0,'trg-planner_amd')
import trg_planner
from trg_planner import synth
cloud = synth.mountain_tile(0, 1000, 0, 1000, seed=99)
for S in (48, 64):
    prm = dict(expand_dist=0.6, robot_size=0.3, sample_num=S, height_threshold=0.16, collision_threshold=0.1, update_collision_threshold=0.5, safety_factor=3.0, goal_tolerance=0.8)
    gs = {}
    for mode in ("device", "host"):
        e = trg_planner.Engine(**prm); e.set_sampler(3, 16); e.set_option("replay", mode); e.set_global_map(cloud)
        import time; t=time.time(); e.init_graph([50.,50.,0.]); dt=time.time()-t
        st=e.stats(); g=e.graph("global"); gs[mode]=g
        print(S, mode, g.V, g.E, f"{dt*1e3:.1f} ms", "used_device", st["used_device_bfs"], "spin", st["bfs_max_spin"], "fallbacks", st["bfs_fallbacks"], e.fallback_reason, flush=True)
    a,b=gs["device"],gs["host"]
    ok = a.V==b.V and a.E==b.E and np.array_equal(a.col,b.col) and np.array_equal(a.xyz.view(np.uint32),b.xyz.view(np.uint32)) and np.array_equal(a.w.view(np.uint32),b.w.view(np.uint32))
    print("S",S,"device==host:",ok, flush=True)
