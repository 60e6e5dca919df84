"""CPU: the compiled pybind11 module exposes the reference's Python surface for the graph engine
(python/trg_planner/pybind/trg_planner_pybind.cpp:19-43), and the C++ shim under it
(include/trg_shim.hpp) compiles, links and refuses to run without a GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PYBIND = "/root/reference/python/trg_planner/pybind/trg_planner_pybind.cpp"

# the surface of trg_planner_pybind.cpp, by line of that file
EXPECTED = {
    "TRG": ["getGraphCopy"],                                   # :23-25
    "Edge": ["dst_id", "weight", "dist"],                      # :27-31
    "NodeState": ["Valid", "Invalid", "Frontier"],             # :33-36
    "Node": ["id", "pos", "state", "edges"],                   # :38-43
    "TRGPlanner": ["init", "setParams", "getTRG", "setPose", "setObs", "setGoal", "getPlannedPath",
                   "getPathInfo", "getMapEigen", "getGoalPose", "getGoalQuat", "shutdown",
                   "setFlagPathFound", "getFlagPreMap", "getFlagPathFound", "getFlagGoalIn",
                   "getFlagGraphInit"],                        # :45-77
}


def test_module_is_compiled_and_has_the_reference_surface():
    import trg_planner
    from trg_planner import _trg_pybind
    assert _trg_pybind.__file__.endswith(".so")
    assert trg_planner.__version__ == "1.0.0" and _trg_pybind.__version__ == "1.0.0"
    for cls in ("TRG", "Edge", "NodeState", "Node"):
        assert getattr(trg_planner, cls) is getattr(_trg_pybind, cls)
    for cls, names in EXPECTED.items():
        c = getattr(trg_planner, cls)
        for n in names:
            assert hasattr(c, n), (cls, n)
    if os.path.exists(REF_PYBIND):  # the list above IS the reference's (checked where it is present)
        src = open(REF_PYBIND).read()
        blocks = re.split(r"py::(?:class_|enum_)<", src)[1:]
        found = {}
        for b in blocks:
            name = re.search(r'\(m,\s*"(\w+)"\)', b).group(1)
            found[name] = re.findall(r'\.(?:def|def_readwrite|value)\("(\w+)"', b)
        assert found == EXPECTED, found


def test_edge_node_semantics():
    import trg_planner as t
    e = t.Edge(7, 0.25, 0.5)
    assert (e.dst_id, e.weight, e.dist) == (7, 0.25, 0.5)
    e.weight = 0.125
    assert e.weight == 0.125
    assert int(t.NodeState.Valid) == 0 and int(t.NodeState.Invalid) == -1 and int(t.NodeState.Frontier) == 1
    n = t.Node(3, np.array([1.0, 2.0], np.float32), 0.5, t.NodeState.Frontier)
    assert n.id == 3 and n.state == t.NodeState.Frontier and n.edges == []
    assert n.pos.dtype == np.float32 and n.pos.shape == (3,) and n.pos.tolist() == [1.0, 2.0, 0.5]
    n.pos[2] = 9.0          # a writable view of the node's own Vector3f, like pybind11/eigen.h gives
    assert n.pos.tolist() == [1.0, 2.0, 9.0]
    n.pos = [4, 5, 6]
    assert n.pos.tolist() == [4.0, 5.0, 6.0]
    n.edges = [e]
    n.edges[0].weight = 0.75  # edges are shared objects (the reference holds Edge*)
    assert e.weight == 0.75
    n.state = t.NodeState.Invalid
    assert n.state == t.NodeState.Invalid
    with pytest.raises(TypeError):
        t.TRG(False, 0.6)   # the nine constructor arguments of trg.h:51-59 are required


def test_trg_has_no_cpu_fallback():
    import torch
    import trg_planner as t
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError) as ei:
        t.TRG(False, 0.6, 0.3, 7, 0.16, 0.1, 0.5, 3.0, 0.8)
    assert "TRG_ERR_DEVICE" in str(ei.value)


def test_cpp_shim_compiles_and_links(tmp_path):
    import trg_planner
    csrc = os.path.dirname(trg_planner.LIB_PATH)
    exe = tmp_path / "shim_check"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror",
                           os.path.join(ROOT, "tests", "cpp", "shim_check.cpp"),
                           "-L", csrc, "-ltrg_engine", "-Wl,-rpath," + csrc, "-o", str(exe)])
    import torch
    if torch.cuda.is_available():
        pytest.skip("run by tests/test_gpu_cabi_c.py on the GPU box")
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok no-gpu"), out.stdout + out.stderr
