"""Python mirror of the reference's ``TRG`` / ``Node`` / ``Edge`` / ``NodeState`` surface.

Reference: python/trg_planner/pybind/trg_planner_pybind.cpp:23-43 and
cpp/trg_planner/core/trg_planner/include/graph/trg.h:18-98.  Method names, argument order and
meaning follow the reference; Eigen vectors become numpy arrays.
"""
from __future__ import annotations

import enum

import numpy as np

from ._engine import Engine


class NodeState(enum.IntEnum):  # trg.h:27-31
    Valid = 0
    Invalid = -1
    Frontier = 1


class Edge:  # trg.h:20-25 / pybind :26-30
    def __init__(self, dst_id, weight, dist):
        self.dst_id = int(dst_id)
        self.weight = float(weight)
        self.dist = float(dist)

    def __repr__(self):
        return f"Edge(dst_id={self.dst_id}, weight={self.weight:.6g}, dist={self.dist:.6g})"


class Node:  # trg.h:33-40 / pybind :37-43
    def __init__(self, id, pos2d, z, state):
        self.id = int(id)
        self.pos = np.array([pos2d[0], pos2d[1], z], dtype=np.float32)
        self.state = NodeState(int(state))
        self.edges = []

    def __repr__(self):
        return f"Node(id={self.id}, pos={self.pos.tolist()}, state={self.state.name}, deg={len(self.edges)})"


class TRG:
    """reference ``class TRG`` (trg.h:18-147); same constructor arguments, same method names."""

    def __init__(self, isVerbose, expand_dist, robot_size, sample_num, height_threshold,
                 collision_threshold, update_collision_threshold, safety_factor, goal_tolerance,
                 device=0):
        self.engine = Engine(expand_dist=expand_dist, robot_size=robot_size, sample_num=sample_num,
                             height_threshold=height_threshold,
                             collision_threshold=collision_threshold,
                             update_collision_threshold=update_collision_threshold,
                             safety_factor=safety_factor, goal_tolerance=goal_tolerance,
                             is_verbose=isVerbose, device=device)

    # -- sampler: explicit here because the reference seeds from std::random_device (trg.cpp:20)
    def setSampler(self, seed=1, table_bits=16):
        self.engine.set_sampler(seed, table_bits)

    # -- map / graph build
    def setGlobalMap(self, cloud_xyz):
        self.engine.set_global_map(cloud_xyz)

    def voxelFilter(self, cloud_xyz, leaf):
        """The pcl::VoxelGrid step of TRGPlanner::loadPrebuiltMap (trg_planner.cpp:91-94)."""
        return self.engine.voxel_filter(cloud_xyz, leaf)

    def setLocalMap(self, start2d, cloud_xyz):
        self.engine.set_local_map(start2d, cloud_xyz)

    def initGraph(self, isPreMap=True, start3d=(0.0, 0.0, 0.0)):
        self.engine.init_graph(start3d)

    def updateGraph(self):
        self.engine.update_graph()

    def resetGraph(self, type="global"):
        self.engine.reset_graph(type)

    def resetMap(self, type="global"):
        self.engine.reset_map(type)

    def saveGraph(self, filepath):
        self.engine.save_json(filepath)

    def loadPrebuiltGraph(self, filepath):
        self.engine.load_json(filepath)

    # -- queries
    def isCollision(self, pos2d, type="global", threshold=0.1):
        flag, _, _ = self.engine.is_collision(np.asarray(pos2d, np.float32).reshape(1, 2), type,
                                              threshold)
        return bool(flag[0])

    def isFrontier(self, pos2d):
        return bool(self.engine.is_frontier(np.asarray(pos2d, np.float32).reshape(1, 2))[0])

    def planSafePath(self, start2d, goal_pose):
        """Returns (found, out_path[n,3], direct_dist, path_length, avg_risk) -- the reference's
        bool return plus its four reference out-parameters (trg.cpp:603-608)."""
        path, info = self.engine.plan(start2d, goal_pose)
        return info.num_points > 0, path, info.direct_dist, info.path_length, info.avg_risk

    def checkReadched(self, pos2d):  # (sic) the reference's spelling, trg.h:78
        return self.engine.check_reached(pos2d)

    def checkReplan(self, pos2d, path):
        return self.engine.check_replan(pos2d, path)

    def refinePath(self, in_path):
        return self.engine.refine_path(in_path)

    # -- export
    def getGraphCSR(self, type="global"):
        return self.engine.graph(type)

    def getGraphCopy(self, type="global"):
        """dict[int, Node], as the reference's pybind returns (trg.cpp:810-824)."""
        g = self.engine.graph(type)
        ids = g.cid if type == "local" else np.arange(g.V)
        out = {}
        for row in range(g.V):
            n = Node(int(ids[row]), g.xyz[row, :2], g.xyz[row, 2], int(g.state[row]))
            a, b = g.rowptr[row], g.rowptr[row + 1]
            n.edges = [Edge(g.col[k], g.w[k], g.dist[k]) for k in range(a, b)]
            out[n.id] = n
        return out

    getGraph = getGraphCopy

    def lockGraph(self):  # single-threaded entry here; kept for source compatibility
        pass

    def unlockGraph(self):
        pass
