#!/usr/bin/env python3
"""Scan a few sampler / geometry configurations for natural nearest-node ties (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "trg-planner_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa
import trg_planner
from trg_planner import synth
import oracle_api as oa
cloud = synth.mountain_cloud(260, 260, seed=5, amplitude=0.3)
for bits in (2, 3, 4, 5, 6):
    for (d, r) in ((0.6, 0.3), (0.5, 0.37), (0.5, 0.3), (0.75, 0.5), (1.0, 0.5)):
        for start in ([13.0, 13.0, 0.0], [12.5, 13.25, 0.0]):
            prm = dict(oa.MOUNTAIN, sample_num=8, expand_dist=d, robot_size=r)
            e = trg_planner.Engine(**prm)
            e.set_sampler(33, bits)
            e.set_global_map(cloud)
            try:
                e.init_graph(start)
            except Exception as ex:
                print(bits, d, r, start, "ERR", ex)
                continue
            st = e.stats()
            print(bits, d, r, start[:2], "dev", st["used_device_bfs"], "levels", st["bfs_levels"], "V", st["created_nodes"],
                  "nn_ties", st["nn_ties"], "fixups", st["bfs_tie_fixups"], "host_levels", st["bfs_host_levels"], flush=True)
            e.close()
