"""GPU parity of the risk weights against BOTH witnesses of the oracle.

The oracle restates trg.cpp:332-338 (mean / centred rows / covariance of the ellipse gather) twice:
literally in fp32 with plain left-to-right sums (the default), and with the same formula accumulated in
fp64 (`set_cov_f64`).  The engine accumulates the covariance in fp64, so:

* against the fp64 witness its weights are the SAME FLOATS on every edge of every case tried (the randomised
  soak `scripts/soak_parity.py`: 0 of 13 M directed entries differ, `profiles/r03_soak_parity.txt`);
* against the fp32 restatement they lie within the north-star tolerance of 1e-5 -- except on about one edge
  per million, where the covariance is nearly degenerate and the fp32 sums' own rounding moves the weight by
  1-2e-5.  The reference's Eigen build sums in yet another (packet) order, so it cannot agree with either
  restatement better than that on such an edge; the soak found six of them in 320 random cases, and round 2's
  code has the very same five.

The cases below are two of those five (small: they run in a second) and a plain one.
"""
import numpy as np
import pytest

from conftest import weight_report

pytestmark = pytest.mark.gpu

MOUNTAIN = dict(expand_dist=0.6, robot_size=0.3, height_threshold=0.16, collision_threshold=0.1,
                update_collision_threshold=0.5, safety_factor=3.0, goal_tolerance=0.8)

# (nx, ny, sample_num, sampler bits, amplitude, cloud seed, sampler seed, start offset) -- cases 62 of
# `soak_parity.py 110 1`, 12 of `soak_parity.py 120 3`, and a plain one
CASES = {
    "soak_1_62": (193, 319, 10, 16, 6.0, 261448309, 185000568, (0.08888011180323385, -0.08386593840725087), 1),
    "soak_3_12": (217, 379, 10, 16, 6.0, 1035471218, 1040393219, (0.427535152040059, 1.8822350525304952), 1),
    "plain": (260, 240, 16, 16, 3.0, 77, 5, (0.0, 0.0), 0),
}


def _structure_equal(a, b):
    return (a.V == b.V and a.E == b.E and np.array_equal(a.rowptr, b.rowptr) and np.array_equal(a.col, b.col)
            and np.array_equal(a.state, b.state) and np.array_equal(a.xyz.view(np.uint32), b.xyz.view(np.uint32))
            and np.array_equal(a.dist.view(np.uint32), b.dist.view(np.uint32)))


@pytest.mark.parametrize("name", list(CASES))
def test_weights_against_both_witnesses(oa, synth, name):
    import trg_planner
    nx, ny, S, bits, amp, seed, sseed, off, expect_outliers = CASES[name]
    cloud = synth.mountain_cloud(nx, ny, seed=seed, amplitude=amp)
    start = [nx * 0.05 + off[0], ny * 0.05 + off[1], 0.0]
    prm = dict(MOUNTAIN, sample_num=S)
    e = trg_planner.Engine(**prm)
    e.set_sampler(sseed, bits)
    e.set_global_map(cloud)
    e.init_graph(start)
    assert e.stats()["used_device_bfs"] == 1
    ge = e.graph("global")
    graphs = {}
    for f64 in (False, True):
        o = oa.Oracle(**prm)
        o.set_sampler(sseed, 0, bits)
        o.set_cov_f64(f64)
        o.set_global_map(cloud)
        assert o.init_graph(start)
        graphs[f64] = o.graph(0)
        o.close()
    for g in graphs.values():  # weights feed no decision: both witnesses build the same structure
        assert _structure_equal(ge, g)
    # the fp64 witness: the same floats (not merely close)
    dw64 = np.abs(ge.w.astype(np.float64) - graphs[True].w.astype(np.float64))
    assert int((dw64 != 0).sum()) == 0, (int((dw64 != 0).sum()), float(dw64.max()))
    # the fp32 restatement: within 1e-5, but for the edges where ITS OWN distance from the fp64 witness is
    # above 1e-5 as well (bounded: 3e-5, at most two edges = four directed entries in these cases)
    flips, over, mx = weight_report(ge.w, graphs[False].w, 1e-5)
    flips_o, over_o, mx_o = weight_report(graphs[False].w, graphs[True].w, 1e-5)
    print(f"{name}: E={ge.E}  vs fp32 restatement: {over} entries over 1e-5 (max {mx:.2e}), {flips} clamp flips;  "
          f"fp32 restatement vs fp64 witness: {over_o} over 1e-5 (max {mx_o:.2e})")
    assert flips == 0
    assert over == over_o and mx == mx_o  # the engine's distance IS the restatement's own noise
    assert mx < 3e-5 and over <= 4
    if expect_outliers:
        assert over >= 2  # (if this stops holding the case no longer shows what it is kept for)
    else:
        assert over == 0
    e.close()
