"""Python mirror of the reference's ``TRGPlanner`` orchestrator.

Reference: cpp/trg_planner/core/trg_planner/{include,src}/planner (PL.h:76-213, PL.cpp) and its
pybind surface (python/trg_planner/pybind/trg_planner_pybind.cpp:45-77).  The orchestrator is
control plane (YAML, two FSM threads, a named-pipe command channel) and is not on the accelerated
path; it is mirrored so that scripts written against the reference keep working: same method
names, same flags, same FSM states, same FIFO wire format (three newline-terminated lines: type,
command, filepath -- interface.cpp:60-165).
"""
from __future__ import annotations

import os
import threading
import time

import numpy as np

from . import config as _config
from . import pcd as _pcd
from . import synth as _synth
from .api import TRG


class OperationResponse:  # interface/operation.h:23-26
    def __init__(self, success=False, message=""):
        self.success, self.message = success, message

    def __bool__(self):
        return bool(self.success)


class TRGPlanner:
    GRAPH_STATES = ("INIT", "UPDATE", "EXPAND", "LOAD", "SAVE", "RESET")
    PLANNING_STATES = ("RESET", "PLANNING", "ONGOING")

    def __init__(self, device=0, map_root=None, sampler_seed=None):
        self._device = device
        self._map_root = map_root  # stands in for the reference's compile-time TRG_DIR/../../
        self._seed = sampler_seed
        self.param_ = None
        self.trg_ = None
        self.is_running = False
        self._graph_state, self._planning_state = "INIT", "RESET"
        self.flag_ = dict(poseIn=False, obsIn=False, goalIn=False, graphInit=False, pathFound=False,
                          planningCount=0)
        self.state_ = dict(frame_id="map", pose3d=np.zeros(3, np.float32),
                           pose2d=np.zeros(2, np.float32), quat=np.array([1, 0, 0, 0], np.float32))
        self.goal_state_ = dict(pose=np.zeros(3, np.float32), quat=np.zeros(4, np.float32), init=False)
        self.cs_ = dict(preMapPtr=None, obsPtr=np.zeros((0, 3), np.float32))
        self.path_ = self._empty_path()
        self._mtx = threading.Lock()
        self._threads = []
        self._fifo_path = None
        self.hz = {}

    @staticmethod
    def _empty_path():
        return dict(raw=np.zeros((0, 3), np.float32), smooth=np.zeros((0, 3), np.float32),
                    direct_dist=0.0, raw_path_length=0.0, smooth_path_length=0.0, planning_time=0.0,
                    avg_risk=0.0)

    # ---- PL.cpp:103-129 -------------------------------------------------------------------
    def setParams(self, config_path):
        self.param_ = _config.load_params(config_path)
        if self._map_root is None:
            self._map_root = os.path.dirname(os.path.dirname(os.path.abspath(config_path)))

    # ---- PL.cpp:21-74 ---------------------------------------------------------------------
    def init(self, start_threads=True, command_interface=False):
        p = self.param_
        self.trg_ = TRG(p.isVerbose, p.expandDist, p.robotSize, p.sampleNum, p.heightThreshold,
                        p.collisionThreshold, p.updateCollisionThreshold, p.safetyFactor,
                        p.goal_tolerance, device=self._device)
        if p.isPreMap:
            self.loadPrebuiltMap()
            self.trg_.setGlobalMap(self.cs_["preMapPtr"])
        self._graph_state = "INIT"
        self.is_running = True
        if start_threads:
            for fn in (self.runGraphFSM, self.runPlanningFSM):
                t = threading.Thread(target=fn, daemon=True)
                t.start()
                self._threads.append(t)
        if command_interface:
            self.setupCommandInterface()

    # ---- PL.cpp:76-101 --------------------------------------------------------------------
    def loadPrebuiltMap(self):
        p = self.param_
        if not p.preMapPath.endswith(".pcd"):
            raise ValueError("Unsupported map type")
        path = p.preMapPath if os.path.isabs(p.preMapPath) else os.path.join(self._map_root, p.preMapPath)
        raw = _pcd.read_pcd(path)
        self.cs_["preMapPtr"] = self.trg_.voxelFilter(raw, p.VoxelSize) if p.isVoxelize else raw

    # ---- FSMs, PL.cpp:131-312 ---------------------------------------------------------------
    def _graph_step(self):
        st = self._graph_state
        if st == "INIT":
            self._graph_state = "UPDATE" if self.flag_["graphInit"] else "INIT"
        elif st == "UPDATE":
            if self.param_.isUpdate and self.flag_["poseIn"] and self.flag_["obsIn"]:
                with self._mtx:
                    obs = self.cs_["obsPtr"]
                    self.trg_.setLocalMap(self.state_["pose2d"], obs)
                    if not self.param_.isPreMap:
                        self.trg_.setGlobalMap(obs)
                self.trg_.updateGraph()
            self._graph_state = "UPDATE"
        elif st == "EXPAND":
            if self._seed is None:  # the reference seeds from std::random_device (trg.cpp:20)
                self.trg_.setSampler(int.from_bytes(os.urandom(4), "little"), 16)
            else:
                self.trg_.setSampler(self._seed, 16)
            self.trg_.initGraph(False, self.state_["pose3d"])
            self.flag_["graphInit"] = True
            self._graph_state = "UPDATE"
        elif st == "LOAD":
            self.trg_.loadPrebuiltGraph(self._abs(self.param_.preGraphPath))
            self.flag_["graphInit"] = True
            self._graph_state = "UPDATE"
        elif st == "RESET":
            self.trg_.resetGraph("global")
            self.trg_.resetGraph("local")
            self.flag_["graphInit"] = False
            self._graph_state = "INIT"
        elif st == "SAVE":
            if self.param_.preGraphPath:
                self.trg_.saveGraph(self._abs(self.param_.preGraphPath))
            self._graph_state = "UPDATE"  # the reference stays in SAVE and re-saves every tick

    def _planning_step(self):
        if self.flag_["goalIn"]:
            self.flag_["goalIn"] = False
            self._planning_state = "PLANNING"
        st = self._planning_state
        if st == "RESET":
            self.flag_["pathFound"] = False
            self.path_ = self._empty_path()
        elif st == "PLANNING":
            self.path_ = self._empty_path()
            t0 = time.perf_counter()
            found, raw, direct, length, risk = self.trg_.planSafePath(self.state_["pose2d"],
                                                                      self.goal_state_["pose"])
            if found:
                self.path_.update(raw=raw, direct_dist=direct, raw_path_length=length, avg_risk=risk,
                                  planning_time=1e3 * (time.perf_counter() - t0),
                                  smooth=self.trg_.refinePath(raw))
                self.flag_["pathFound"] = True
                self.flag_["planningCount"] = 0
                self._planning_state = "ONGOING"
            else:
                self.flag_["planningCount"] += 1
                self._planning_state = "RESET" if self.flag_["planningCount"] > 10 else "PLANNING"
        elif st == "ONGOING":
            if self.trg_.checkReadched(self.state_["pose2d"]):
                self._planning_state = "RESET"
            elif self.trg_.checkReplan(self.state_["pose2d"], self.path_["raw"]):
                self._planning_state = "PLANNING"

    def _loop(self, step, rate, name):
        while self.is_running:
            t0 = time.perf_counter()
            step()
            remain = 1.0 / max(rate, 1e-3) - (time.perf_counter() - t0)
            if remain > 0:
                time.sleep(remain)
            self.hz[name] = round(1.0 / max(time.perf_counter() - t0, 1e-9), 2)

    def runGraphFSM(self):
        self._loop(self._graph_step, self.param_.graph_rate, "graph")

    def runPlanningFSM(self):
        self._loop(self._planning_step, self.param_.planning_rate, "planning")

    def _abs(self, path):
        return path if os.path.isabs(path) else os.path.join(self._map_root, path)

    # ---- setters / getters, PL.cpp:342-452 ----------------------------------------------------
    def getTRG(self):
        return self.trg_

    def setPose(self, pose=(0.0, 0.0, 0.0), quat=(1.0, 0.0, 0.0, 0.0), frame_id="map"):
        with self._mtx:
            self.state_.update(frame_id=frame_id, pose3d=np.asarray(pose, np.float32),
                               pose2d=np.asarray(pose, np.float32)[:2],
                               quat=np.asarray(quat, np.float32))
            self.flag_["poseIn"] = True

    def setObs(self, obs):
        if not self.flag_["poseIn"]:
            print("Pose is not initialized")
            return
        with self._mtx:
            self.cs_["obsPtr"] = np.ascontiguousarray(obs, np.float32).reshape(-1, 3).copy()
            self.flag_["obsIn"] = True

    def setGoal(self, pose=(0.0, 0.0, 0.0), quat=(1.0, 0.0, 0.0, 0.0)):
        if not self.flag_["graphInit"]:
            print("Graph is not initialized")
            return
        with self._mtx:
            self.goal_state_.update(pose=np.asarray(pose, np.float32), quat=np.asarray(quat, np.float32),
                                    init=True)
            self.flag_["goalIn"] = True

    def getPlannedPath(self, type="smooth"):
        if not self.flag_["pathFound"]:
            print("Path is not found")
            return []
        if type not in ("raw", "smooth"):
            print("Invalid path type")
            return []
        return [np.array(p, np.float32) for p in self.path_[type]]

    def getPathInfo(self):
        if not self.flag_["pathFound"]:
            print("Path is not found")
            return []
        p = self.path_
        return [p["direct_dist"], p["raw_path_length"], p["smooth_path_length"], p["planning_time"],
                p["avg_risk"]]

    def getMapEigen(self, type="preMap"):  # the reference's default is an invalid type, too
        if type == "pre":
            if not self.param_.isPreMap:
                print("Prebuilt map is not loaded")
                return np.zeros((0, 3), np.float32)
            return self.cs_["preMapPtr"]
        if type == "obs":
            return self.cs_["obsPtr"]
        print("Invalid map type")
        return np.zeros((0, 3), np.float32)

    def getGoalPose(self):
        return self.goal_state_["pose"] if self.goal_state_["init"] else np.zeros(3, np.float32)

    def getGoalQuat(self):
        return self.goal_state_["quat"] if self.goal_state_["init"] else np.zeros(4, np.float32)

    def shutdown(self):
        self.is_running = False
        if self._fifo_path:
            self.stopCommandInterface()
        for t in self._threads:
            t.join(timeout=5)
        self._threads = []

    def setFlagPathFound(self, flag):
        self.flag_["pathFound"] = bool(flag)

    def getFlagPreMap(self):
        return bool(self.param_.isPreMap)

    def getFlagPathFound(self):
        return self.flag_["pathFound"]

    def getFlagGoalIn(self):
        return self.flag_["goalIn"]

    def getFlagGraphInit(self):
        return self.flag_["graphInit"]

    # ---- command channel, PL.cpp:456-558 / interface.cpp ---------------------------------------
    def processOperation(self, type, command, filepath=""):
        r = OperationResponse()
        if type == "graph":
            if command == "expand":
                self._graph_state, r.success, r.message = "EXPAND", True, "Graph expansion triggered"
            elif command == "load":
                if not filepath:
                    r.message = "Filepath is required for load operation"
                    return r
                self.param_.preGraphPath = filepath
                self._graph_state, r.success = "LOAD", True
                r.message = "Graph load triggered with path: " + filepath
            elif command == "reset":
                self._graph_state, r.success, r.message = "RESET", True, "Graph reset triggered"
            elif command == "save":
                if not filepath:
                    r.message = "Filepath is required for save operation"
                    return r
                self._graph_state, r.success = "SAVE", True
                r.message = "Graph save triggered with path: " + filepath
            else:
                r.message = "Invalid graph command: " + command
        elif type == "path":
            if command == "plan":
                if not self.flag_["graphInit"]:
                    r.message = "Graph is not initialized"
                    return r
                self.flag_["goalIn"], r.success, r.message = True, True, "Path planning triggered"
            elif command == "reset":
                self._planning_state, r.success, r.message = "RESET", True, "Path reset triggered"
            elif command in ("load", "save"):
                r.message = ("Filepath is required for " + command + " operation") if not filepath \
                    else ("Path " + command + " operation not implemented yet")
            else:
                r.message = "Invalid path command: " + command
        else:
            r.message = "Invalid operation type. Must be 'graph' or 'path'."
        return r

    def setupCommandInterface(self, pipe_path="/tmp/trg_planner_fifo"):
        """Named-pipe listener (interface.cpp:19-29, 60-131): the pipe is opened non-blocking, switched
        back to blocking, and ONE request -- three newline-terminated lines type / command / filepath
        -- is read per open; an empty read (no writer) is retried after 100 ms."""
        import fcntl
        if os.path.exists(pipe_path):
            os.unlink(pipe_path)
        os.mkfifo(pipe_path, 0o666)
        self._fifo_path = pipe_path
        self._fifo_running = True

        def listen():
            while self._fifo_running:
                try:
                    fd = os.open(pipe_path, os.O_RDONLY | os.O_NONBLOCK)
                except OSError:
                    time.sleep(1.0)
                    continue
                fcntl.fcntl(fd, fcntl.F_SETFL, fcntl.fcntl(fd, fcntl.F_GETFL) & ~os.O_NONBLOCK)
                with os.fdopen(fd, "r") as f:
                    req = [f.readline()[:1023].rstrip("\n") for _ in range(3)]
                if req[0] and req[1]:
                    self.processOperation(req[0], req[1], req[2])
                time.sleep(0.1)

        t = threading.Thread(target=listen, daemon=True)
        t.start()
        self._fifo_thread = t
        return True

    def stopCommandInterface(self):
        """TRGInterface::stopCommandListener + cleanup (interface.cpp:47-58, 31-33)."""
        self._fifo_running = False
        t = getattr(self, "_fifo_thread", None)
        if t is not None:
            t.join(timeout=3)
        if self._fifo_path and os.path.exists(self._fifo_path):
            os.unlink(self._fifo_path)
        self._fifo_path = None

    @staticmethod
    def sendCommand(type, command, filepath="", pipe_path="/tmp/trg_planner_fifo"):
        """TRGInterface::sendCommand (interface.cpp:133-165): three lines into the named pipe."""
        import stat
        r = OperationResponse()
        try:
            is_fifo = stat.S_ISFIFO(os.stat(pipe_path).st_mode)
        except OSError:
            is_fifo = False
        if not is_fifo:
            r.message = "Command pipe not found. Is TRG Planner running?"
            return r
        try:
            with open(pipe_path, "w") as f:
                f.write(f"{type}\n{command}\n{filepath}\n")
        except OSError:
            r.message = "Failed to open command pipe. Is TRG Planner running?"
            return r
        r.success = True
        r.message = "Command sent: " + type + " " + command + (f" (file: {filepath})" if filepath else "")
        return r
