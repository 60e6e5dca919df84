#!/usr/bin/env python3
"""Randomised parity soak (GPU box): engine vs CPU oracle over random terrains, sample counts, sampler
seeds / direction counts and starts; prints one line per case and a summary."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "trg-planner_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import numpy as np
import trg_planner
from trg_planner import synth
import oracle_api as oa
from conftest import assert_graph_equal
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t_all = time.time()
for case in range(n_cases):
    nx, ny = int(rng.integers(150, 420)), int(rng.integers(150, 420))
    S = int(rng.choice([5, 7, 10, 16, 24]))
    bits = int(rng.choice([16, 16, 16, 12, 5, 4]))
    amp = float(rng.choice([0.3, 1.5, 3.0, 6.0]))
    seed, sseed = int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1 << 30))
    cloud = synth.mountain_cloud(nx, ny, seed=seed, amplitude=amp)
    start = [nx * 0.05 + float(rng.uniform(-2, 2)), ny * 0.05 + float(rng.uniform(-2, 2)), 0.0]
    prm = dict(oa.MOUNTAIN, sample_num=S)
    e = trg_planner.Engine(**prm)
    e.set_sampler(sseed, bits)
    e.set_option("keep_preclean", 1)
    o = oa.Oracle(**prm)
    o.set_sampler(sseed, 0, bits)
    o.set_global_map(cloud)
    ok_o = o.init_graph(start)
    try:
        e.set_global_map(cloud)
        e.init_graph(start)
        ok_e = True
    except trg_planner.TrgError as ex:
        ok_e = False
        msg = str(ex)
    line = f"case {case}: {nx}x{ny} S={S} bits={bits} amp={amp} "
    if ok_o != ok_e:
        bad += 1
        print(line + f"ROOT MISMATCH oracle={ok_o} engine={ok_e}", flush=True)
        continue
    if not ok_o:
        print(line + "no root (both)", flush=True)
        continue
    st = e.stats()
    try:
        assert_graph_equal(e.graph("preclean"), o.graph(1), 1e-5)
        assert_graph_equal(e.graph("global"), o.graph(0), 1e-5)
        c = o.counters()
        assert st["trials"] == c["trials"] and st["samples"] == c["samples"], (st["trials"], c["trials"])
        res = "ok"
    except AssertionError as ex:
        bad += 1
        res = "MISMATCH " + str(ex)[:200]
    g = e.graph("global")
    print(line + f"V={g.V} E={g.E} dev={st['used_device_bfs']} fallbacks={st['bfs_fallbacks']} host_levels={st['bfs_host_levels']} "
          f"tie_fixups={st['bfs_tie_fixups']} nn_ties={st['nn_ties']} map_ties={st['map_nn_resolved']} {res}", flush=True)
    e.close()
    o.close()
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t_all:.1f} s")
