"""The N > 1 exchange code over the real backend ("nccl" = RCCL) with ONE rank: the multi-GPU scaling
run belongs to the driver (no multi-GPU node is available to this build), so this at least runs the very
calls bench.py --gpus N makes -- process group on the GPU, the engine's own RCCL communicator (unique id over
torch.distributed) and trg_engine_stitch_exchange, and the torch.distributed variant of the two exchanges
(all-gather of device tensors with ragged sizes, the native stitch fed from their results) -- on the hardware
and backend they will run on."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "trg-planner_amd"))
import numpy as np
import torch
import torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = "29531"
os.environ["TRG_FORCE_COLLECTIVES"] = "1"
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import trg_planner
from trg_planner import synth, tiled, tiling
dev = torch.device("cuda", 0)
a = torch.arange(15, dtype=torch.int32, device=dev).reshape(5, 3)
cat, counts = tiled.allgatherv_t(a, dist)
assert counts == [5] and torch.equal(cat, a) and cat.is_cuda
e0, c0 = tiled.allgatherv_t(torch.zeros((0, 6), dtype=torch.int32, device=dev), dist)
assert c0 == [0] and e0.shape == (0, 6)
items, secs = tiling.reduce_throughput(1234.0, 0.5, dist, dev)
assert items == 1234.0 and secs == 0.5
MOUNTAIN = dict(expand_dist=0.6, robot_size=0.3, height_threshold=0.16, collision_threshold=0.1,
                update_collision_threshold=0.5, safety_factor=3.0, goal_tolerance=0.8, sample_num=8)
cloud = synth.mountain_tile(0, 300, 0, 300, seed=3)
eng = trg_planner.Engine(**MOUNTAIN, device=0)
eng.set_sampler(7, 16)
core = tiled.tile_cores(1, 1, 300, 300)[0]
eng.set_tile(core, epoch=0)
eng.set_global_map_device(torch.from_numpy(cloud).to(dev).data_ptr(), cloud.shape[0], 3)
eng.init_graph([15.0, 15.0, 0.0])
# the exchanges inside the engine (trg_engine_stitch_exchange on the engine's own RCCL communicator, the
# unique id carried by torch.distributed) ...
info = tiled.stitch_device(eng, 0, core, 1, 1, dist)
V, E = eng.graph_sizes("stitched")
assert info["backend"] == "rccl-native" and info["n_cross"] == 0, info
assert (V, E) == tuple(eng.graph_sizes("global")), (V, E)
g_native = eng.graph("stitched")
# ... and in torch.distributed (device tensors over the "nccl" backend): the same rows
os.environ["TRG_NATIVE_EXCHANGE"] = "0"
info = tiled.stitch_device(eng, 0, core, 1, 1, dist)
assert info["backend"] == "nccl" and info["n_cross"] == 0, info
g_torch = eng.graph("stitched")
for k in ("rowptr", "col", "state", "cid"):
    assert np.array_equal(getattr(g_native, k), getattr(g_torch, k)), k
assert np.array_equal(g_native.w.view(np.uint32), g_torch.w.view(np.uint32))
eng.comm_destroy()
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK", V, E)
"""


def test_exchange_calls_over_rccl_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
