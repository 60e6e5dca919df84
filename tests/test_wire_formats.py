"""CPU: the wire formats either side of the hot path.

* graph JSON (TRG::saveGraph / loadPrebuiltGraph, trg.cpp:66-177) pinned against nlohmann/json, the
  library the reference writes and reads it with (3.1.1 ships in this image);
* the FIFO command channel (interface.cpp:60-165, operation.h:6-26): three newline-terminated lines
  type / command / filepath written into a named pipe.
"""
import os
import subprocess
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NLOHMANN_DIR = "/opt/conda/include"


@pytest.mark.skipif(not os.path.exists(os.path.join(NLOHMANN_DIR, "json.hpp")),
                    reason="nlohmann/json header not in this image")
@pytest.mark.parametrize("sanitize", [False, True])
def test_graph_json_against_nlohmann(tmp_path, sanitize):
    exe = tmp_path / "graph_json_check"
    flags = ["-O1", "-g", "-std=c++17"]
    if sanitize:
        flags += ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
    subprocess.check_call(["g++"] + flags + ["-idirafter", NLOHMANN_DIR,
                                             os.path.join(ROOT, "tests", "cpp", "graph_json_check.cpp"),
                                             "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def _write_request(pipe, type_, command, filepath=""):
    """What TRGInterface::sendCommand puts on the wire (interface.cpp:153-156): three lines, each
    terminated by std::endl, i.e. '\\n' + flush -- three separate writes."""
    fd = os.open(pipe, os.O_WRONLY)
    try:
        for part in (type_, command, filepath):
            os.write(fd, (part + "\n").encode())
    finally:
        os.close(fd)


def test_fifo_command_channel(tmp_path):
    import trg_planner
    pipe = str(tmp_path / "trg_planner_fifo")
    pl = trg_planner.TRGPlanner()
    pl.param_ = trg_planner.config.Params({})  # every key at its trg_planner.cpp:108-128 default
    seen = []
    inner = pl.processOperation

    def spy(type, command, filepath=""):
        r = inner(type, command, filepath)
        seen.append((type, command, filepath, r.success, r.message))
        return r

    pl.processOperation = spy
    # no pipe yet: the sender reports it exactly like the reference
    r = trg_planner.TRGPlanner.sendCommand("graph", "expand", "", pipe_path=pipe)
    assert not r.success and r.message == "Command pipe not found. Is TRG Planner running?"
    assert pl.setupCommandInterface(pipe)
    try:
        import stat
        assert stat.S_ISFIFO(os.stat(pipe).st_mode)

        def wait_for(n):
            t0 = time.time()
            while len(seen) < n and time.time() - t0 < 10:
                time.sleep(0.02)
            assert len(seen) >= n, seen

        _write_request(pipe, "graph", "expand")
        wait_for(1)
        assert seen[-1][:4] == ("graph", "expand", "", True) and pl._graph_state == "EXPAND"
        _write_request(pipe, "graph", "save", "graphs/g.json")
        wait_for(2)
        assert seen[-1][:4] == ("graph", "save", "graphs/g.json", True) and pl._graph_state == "SAVE"
        assert seen[-1][4] == "Graph save triggered with path: graphs/g.json"
        _write_request(pipe, "graph", "load", "graphs/g.json")
        wait_for(3)
        assert seen[-1][3] and pl._graph_state == "LOAD" and pl.param_.preGraphPath == "graphs/g.json"
        _write_request(pipe, "graph", "reset")
        wait_for(4)
        assert seen[-1][3] and pl._graph_state == "RESET"
        # path plan needs an initialised graph (trg_planner.cpp:521-525)
        _write_request(pipe, "path", "plan")
        wait_for(5)
        assert not seen[-1][3] and seen[-1][4] == "Graph is not initialized"
        pl.flag_["graphInit"] = True
        _write_request(pipe, "path", "plan")
        wait_for(6)
        assert seen[-1][3] and pl.flag_["goalIn"] is True
        pl._planning_state = "PLANNING"
        _write_request(pipe, "path", "reset")
        wait_for(7)
        assert seen[-1][3] and pl._planning_state == "RESET"
        # refusals: missing filepath, unknown command / type
        _write_request(pipe, "graph", "load")
        wait_for(8)
        assert not seen[-1][3] and seen[-1][4] == "Filepath is required for load operation"
        _write_request(pipe, "graph", "frobnicate")
        wait_for(9)
        assert seen[-1][4] == "Invalid graph command: frobnicate"
        _write_request(pipe, "map", "expand")
        wait_for(10)
        assert seen[-1][4] == "Invalid operation type. Must be 'graph' or 'path'."
        # the package's own sender writes the same three lines
        r = trg_planner.TRGPlanner.sendCommand("graph", "expand", "", pipe_path=pipe)
        assert r.success and r.message == "Command sent: graph expand"
        wait_for(11)
        assert seen[-1][:3] == ("graph", "expand", "")
        # a request without a command line is ignored (interface.cpp:108)
        fd = os.open(pipe, os.O_WRONLY)
        os.write(fd, b"graph\n")
        os.close(fd)
        time.sleep(0.4)
        assert len(seen) == 11
    finally:
        pl.stopCommandInterface()
    assert not os.path.exists(pipe)
