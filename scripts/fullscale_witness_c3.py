#!/usr/bin/env python3
"""Full-size run (BASELINE config 3: 10 M points, S=16) against the oracle's SECOND witness: the same
restatement of trg.cpp with the covariance of the ellipse gather accumulated in fp64 (`set_cov_f64`).
The engine accumulates in fp64 as well, so its weights should be the same floats, edge for edge; the
script reports how many of the directed entries differ at all, writes `gpurun_out/fullscale_witness_c3.json`
(copied to profiles/) and the SHA-256 of the witness's weight array that tests/test_gpu_c3_fullsize.py
checks every later engine build against (tests/golden/c3_witness_digest.json).

usage: python scripts/fullscale_witness_c3.py [nx ny]   (default 3200 3125)
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "trg-planner_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: F401,E402
import oracle_api as oa  # noqa: E402
import trg_planner  # noqa: E402
from trg_planner import synth  # noqa: E402

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3200, 3125)
prm = dict(oa.MOUNTAIN, sample_num=16)
cloud = synth.mountain_tile(0, nx, 0, ny, seed=20250418)
start = [nx * 0.05, ny * 0.05, 0.0]
e = trg_planner.Engine(**prm)
e.set_sampler(7, 16)
e.set_global_map(cloud)
e.init_graph(start)
ge = e.graph("global")
print(f"engine: V'={ge.V} E'={ge.E}", flush=True)

oa.use_reference_kd(True)
o = oa.Oracle(**prm)
o.set_sampler(7, 0, 16)
o.set_cov_f64(True)
t0 = time.time()
o.set_global_map(cloud)
assert o.init_graph(start)
t_oracle = time.time() - t0
go = o.graph(0)
print(f"fp64 witness: V'={go.V} E'={go.E} in {t_oracle:.1f}s", flush=True)

same = (ge.V == go.V and ge.E == go.E and np.array_equal(ge.rowptr, go.rowptr) and np.array_equal(ge.col, go.col)
        and np.array_equal(ge.state, go.state) and np.array_equal(ge.xyz.view(np.uint32), go.xyz.view(np.uint32))
        and np.array_equal(ge.dist.view(np.uint32), go.dist.view(np.uint32)))
res = {"nx": nx, "ny": ny, "points": int(cloud.shape[0]), "V": go.V, "E": go.E, "structure_bit_equal": bool(same),
       "witness_s": t_oracle}
if same:
    dw = np.abs(ge.w.astype(np.float64) - go.w.astype(np.float64))
    res["entries_that_differ_at_all"] = int((dw != 0).sum())
    res["weight_max_abs_diff"] = float(dw.max())
    res["witness_w_sha256"] = hashlib.sha256(np.ascontiguousarray(go.w, np.float32).tobytes()).hexdigest()
    res["engine_w_sha256"] = hashlib.sha256(np.ascontiguousarray(ge.w, np.float32).tobytes()).hexdigest()
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "fullscale_witness_c3.json"), "w"), indent=1)
if same and (nx, ny) == (3200, 3125):
    json.dump({"V": go.V, "E": go.E, "w_sha256_fp64_witness": res["witness_w_sha256"],
               "made_by": "scripts/fullscale_witness_c3.py (the ORACLE's graph with set_cov_f64)"},
              open(os.path.join(ROOT, "gpurun_out", "c3_witness_digest.json"), "w"), indent=1)
