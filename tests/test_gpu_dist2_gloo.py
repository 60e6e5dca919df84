"""GPU, world_size 2 on ONE card: the product path of the multi-rank build -- every rank builds its tile on
the GPU and runs tiled.stitch_device (native boundary / cross / assemble steps) -- with the two exchanges
carried as host tensors over gloo (RCCL refuses two ranks on one device; on a multi-GPU node the same call
runs them inside the engine over RCCL).  Rank 0 collects both ranks' rows and compares the assembled global
graph with the tiled CPU oracle."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
root = %(root)r
sys.path.insert(0, os.path.join(root, "trg-planner_amd"))
sys.path.insert(0, os.path.join(root, "oracle"))
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), 2
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
import trg_planner
from trg_planner import synth, tiled
cols, rows, nx, ny, halo, seed, sseed = 2, 1, 130, 130, 11, 77, 9
prm = dict(expand_dist=0.6, robot_size=0.3, sample_num=7, height_threshold=0.16, collision_threshold=0.1,
           update_collision_threshold=0.5, safety_factor=3.0, goal_tolerance=0.8)
terrain = dict(amplitude=2.0, wavelength=20.0)
cores = tiled.tile_cores(cols, rows, nx, ny)
core = cores[rank]
cloud = synth.mountain_tile(*tiled.tile_lattice_window(rank, cols, rows, nx, ny, halo), seed=seed, **terrain)
e = trg_planner.Engine(**prm)
e.set_sampler(sseed, 16)
e.set_tile(core, epoch=rank)
e.set_global_map(cloud)
e.init_graph([0.5 * (core[0] + core[2]), 0.5 * (core[1] + core[3]), 0.0])
info = tiled.stitch_device(e, rank, core, cols, rows, dist, torch.device("cpu"))
assert info["backend"] == "gloo" and info["n_cross"] > 10, info
g = e.graph("stitched")
mine = dict(rowptr=g.rowptr, col=g.col, w=g.w, dist=g.dist, xyz=g.xyz, state=g.state, cid=g.cid)
parts = [None, None] if rank == 0 else None
dist.gather_object(mine, parts, dst=0)
if rank == 0:
    import oracle_api as oa
    import tiled_oracle
    oa.build()
    class P:
        def __init__(s, d): s.__dict__.update(d); s.V = d["state"].shape[0]
    G = tiled.concat_stitched([P(p) for p in parts])
    _, o_st, OG = tiled_oracle.build_tiled_oracle(oa, synth, tiled, prm, cols, rows, nx, ny, halo, seed, sseed, terrain)
    assert info["n_cross"] == o_st[0].shape[0], (info, o_st[0].shape)
    assert G["V"] == OG["V"] and np.array_equal(G["rowptr"], OG["rowptr"]) and np.array_equal(G["col"], OG["col"])
    assert np.array_equal(G["state"], OG["state"])
    assert np.array_equal(G["xyz"].view(np.uint32), OG["xyz"].view(np.uint32))
    assert np.array_equal(G["dist"].view(np.uint32), OG["dist"].view(np.uint32))
    assert float(np.abs(G["w"] - OG["w"]).max()) <= 1e-5
    print("DIST2_OK", G["V"], int(G["col"].size), info["n_cross"])
dist.barrier()
dist.destroy_process_group()
"""


def test_two_ranks_one_gpu_stitch_device_against_tiled_oracle():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD % {"root": ROOT}], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-2000:], se[-4000:])
    assert "DIST2_OK" in outs[0][0], outs[0]
