"""ctypes wrapper of oracle/_build/libtrg_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (trg-planner_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libtrg_oracle.so")
REF_KD = os.path.join(HERE, "_ref", "libkdtree_ref.so")

COUNTER_NAMES = [
    "collision_queries", "collision_hits", "nn_map_queries", "ellipse_queries", "ellipse_hits",
    "wire_calls", "wire_evals", "wire_ok", "wire_gate", "wire_seg", "wire_empty", "wire_few",
    "wire_clamped", "expanded", "trials", "samples", "created", "invalid_created",
    "nn_node_queries", "sample_hits", "wire_hits_new", "wire_hits_other",
]


def build(force=False):
    if force or not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        L.trg_oracle_create.restype = C.c_void_p
        L.trg_oracle_create.argtypes = [C.c_float, C.c_float, C.c_int] + [C.c_float] * 5
        L.trg_oracle_destroy.argtypes = [C.c_void_p]
        L.trg_oracle_set_kd_backend.argtypes = [C.c_int, C.c_char_p]
        L.trg_oracle_set_sampler.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int]
        L.trg_oracle_set_cov_f64.argtypes = [C.c_void_p, C.c_int]
        L.trg_oracle_set_tile.argtypes = [C.c_void_p, fp, C.c_uint32]
        L.trg_oracle_set_trace.argtypes = [C.c_void_p, C.c_int]
        L.trg_oracle_get_table.argtypes = [C.c_void_p, fp, fp]
        L.trg_oracle_set_global_map.argtypes = [C.c_void_p, fp, C.c_size_t, C.c_size_t]
        L.trg_oracle_set_local_map.argtypes = [C.c_void_p, C.c_float, C.c_float, fp, C.c_size_t,
                                               C.c_size_t]
        L.trg_oracle_init_graph.argtypes = [C.c_void_p, fp]
        L.trg_oracle_update_graph.argtypes = [C.c_void_p]
        L.trg_oracle_is_collision.argtypes = [C.c_void_p, C.c_int, C.c_float, fp, C.c_size_t, ip,
                                              ip, ip]
        L.trg_oracle_nearest_z.argtypes = [C.c_void_p, C.c_int, fp, C.c_size_t, fp]
        L.trg_oracle_is_frontier.argtypes = [C.c_void_p, fp, C.c_size_t, ip]
        L.trg_oracle_edge_risk.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_size_t, ip, ip, fp, fp]
        L.trg_oracle_graph_sizes.argtypes = [C.c_void_p, C.c_int, ip, ip]
        L.trg_oracle_graph_export.argtypes = [C.c_void_p, C.c_int, fp, ip, ip, ip, fp, fp, ip]
        L.trg_oracle_trace_size.restype = C.c_size_t
        L.trg_oracle_trace_size.argtypes = [C.c_void_p]
        L.trg_oracle_trace_export.argtypes = [C.c_void_p, ip, ip, ip, ip, fp, fp]
        L.trg_oracle_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.trg_oracle_reset_counters.argtypes = [C.c_void_p]
        L.trg_oracle_plan.argtypes = [C.c_void_p, fp, fp, fp, C.c_int, fp]
        L.trg_oracle_refine.argtypes = [fp, C.c_int, fp, C.c_int]
        L.trg_oracle_svd_u3.argtypes = [fp, fp]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def use_reference_kd(on=True):
    """Route every spatial query of subsequently created oracles through the reference kdtree.c
    compiled in place (oracle/_ref).  Returns False when that library is absent."""
    if on:
        if not os.path.exists(REF_KD):
            return False
        return lib().trg_oracle_set_kd_backend(1, REF_KD.encode()) == 0
    lib().trg_oracle_set_kd_backend(0, None)
    return True


MOUNTAIN = dict(expand_dist=0.6, robot_size=0.3, sample_num=7, height_threshold=0.16,
                collision_threshold=0.1, update_collision_threshold=0.5, safety_factor=3.0,
                goal_tolerance=0.8)
INDOOR = dict(expand_dist=0.4, robot_size=0.3, sample_num=15, height_threshold=0.15,
              collision_threshold=0.1, update_collision_threshold=0.1, safety_factor=3.0,
              goal_tolerance=0.8)


class Graph:
    def __init__(self, xyz, state, rowptr, col, w, dist, cid):
        self.xyz, self.state, self.rowptr, self.col = xyz, state, rowptr, col
        self.w, self.dist, self.cid = w, dist, cid

    @property
    def V(self):
        return self.state.shape[0]

    @property
    def E(self):
        return self.col.shape[0]


class Oracle:
    def __init__(self, expand_dist=0.6, robot_size=0.3, sample_num=7, height_threshold=0.16,
                 collision_threshold=0.1, update_collision_threshold=0.5, safety_factor=3.0,
                 goal_tolerance=0.8):
        self.L = lib()
        self.h = self.L.trg_oracle_create(expand_dist, robot_size, sample_num, height_threshold,
                                          collision_threshold, update_collision_threshold,
                                          safety_factor, goal_tolerance)
        self.table_bits = 16
        self._keep = None

    def close(self):
        if self.h:
            self.L.trg_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_sampler(self, seed=1, mode=0, table_bits=16):
        self.table_bits = table_bits
        self.L.trg_oracle_set_sampler(self.h, mode, seed, table_bits)

    def set_tile(self, core_xyxy, epoch=0):
        """Tiled-build extension: node creation restricted to [x0,x1) x [y0,y1), sampler epoch."""
        c = np.ascontiguousarray(core_xyxy, dtype=np.float32)
        self.L.trg_oracle_set_tile(self.h, _f(c), int(epoch))

    def set_cov_f64(self, on):
        self.L.trg_oracle_set_cov_f64(self.h, int(on))

    def set_trace(self, on):
        self.L.trg_oracle_set_trace(self.h, int(on))

    def table(self):
        n = 1 << self.table_bits
        c = np.empty(n, np.float32)
        s = np.empty(n, np.float32)
        self.L.trg_oracle_get_table(self.h, _f(c), _f(s))
        return c, s

    def set_global_map(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        self.L.trg_oracle_set_global_map(self.h, _f(xyz), xyz.shape[0], xyz.shape[1])

    def set_local_map(self, start2d, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        self.L.trg_oracle_set_local_map(self.h, float(start2d[0]), float(start2d[1]), _f(xyz),
                                        xyz.shape[0], xyz.shape[1])

    def init_graph(self, start3d):
        s = np.ascontiguousarray(start3d, dtype=np.float32)
        return self.L.trg_oracle_init_graph(self.h, _f(s)) == 0

    def update_graph(self):
        self.L.trg_oracle_update_graph(self.h)

    def is_collision(self, xy, kind=0, threshold=0.1):
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        m = xy.shape[0]
        flag = np.empty(m, np.int32)
        cnt = np.empty(m, np.int32)
        n = np.empty(m, np.int32)
        self.L.trg_oracle_is_collision(self.h, kind, threshold, _f(xy), m, _i(flag), _i(cnt), _i(n))
        return flag, cnt, n

    def nearest_z(self, xy, kind=0):
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        z = np.empty(xy.shape[0], np.float32)
        self.L.trg_oracle_nearest_z(self.h, kind, _f(xy), xy.shape[0], _f(z))
        return z

    def is_frontier(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        flag = np.empty(xy.shape[0], np.int32)
        self.L.trg_oracle_is_frontier(self.h, _f(xy), xy.shape[0], _i(flag))
        return flag

    def edge_risk(self, p1, p2, kind=0):
        p1 = np.ascontiguousarray(p1, dtype=np.float32)
        p2 = np.ascontiguousarray(p2, dtype=np.float32)
        m = p1.shape[0]
        status = np.empty(m, np.int32)
        n_pts = np.empty(m, np.int32)
        w = np.empty(m, np.float32)
        d = np.empty(m, np.float32)
        self.L.trg_oracle_edge_risk(self.h, kind, _f(p1), _f(p2), m, _i(status), _i(n_pts), _f(w),
                                    _f(d))
        return status, n_pts, w, d

    def graph(self, which=0):
        V = C.c_int()
        E = C.c_int()
        self.L.trg_oracle_graph_sizes(self.h, which, C.byref(V), C.byref(E))
        V, E = V.value, E.value
        xyz = np.empty((V, 3), np.float32)
        state = np.empty(V, np.int32)
        rowptr = np.empty(V + 1, np.int32)
        col = np.empty(E, np.int32)
        w = np.empty(E, np.float32)
        dist = np.empty(E, np.float32)
        cid = np.empty(V, np.int32)
        self.L.trg_oracle_graph_export(self.h, which, _f(xyz), _i(state), _i(rowptr), _i(col),
                                       _f(w), _f(dist), _i(cid))
        return Graph(xyz, state, rowptr, col, w, dist, cid)

    def trace(self):
        n = self.L.trg_oracle_trace_size(self.h)
        src = np.empty(n, np.int32)
        dst = np.empty(n, np.int32)
        status = np.empty(n, np.int32)
        n_pts = np.empty(n, np.int32)
        w = np.empty(n, np.float32)
        d = np.empty(n, np.float32)
        self.L.trg_oracle_trace_export(self.h, _i(src), _i(dst), _i(status), _i(n_pts), _f(w), _f(d))
        return dict(src=src, dst=dst, status=status, n_pts=n_pts, weight=w, dist=d)

    def counters(self):
        out = (C.c_uint64 * len(COUNTER_NAMES))()
        self.L.trg_oracle_counters(self.h, out)
        return dict(zip(COUNTER_NAMES, [int(v) for v in out]))

    def reset_counters(self):
        self.L.trg_oracle_reset_counters(self.h)

    def plan(self, start2d, goal3d, max_pts=100000):
        s = np.ascontiguousarray(start2d, dtype=np.float32)
        g = np.ascontiguousarray(goal3d, dtype=np.float32)
        path = np.empty((max_pts, 3), np.float32)
        info = np.empty(3, np.float32)
        n = self.L.trg_oracle_plan(self.h, _f(s), _f(g), _f(path), max_pts, _f(info))
        return path[:n].copy(), info

    @staticmethod
    def refine(path):
        path = np.ascontiguousarray(path, dtype=np.float32)
        out = np.empty((2 * path.shape[0] + 2, 3), np.float32)
        n = lib().trg_oracle_refine(_f(path), path.shape[0], _f(out), out.shape[0])
        return out[:n].copy()

    @staticmethod
    def svd_u3(A):
        A = np.ascontiguousarray(A, dtype=np.float32)
        U = np.empty((3, 3), np.float32)
        lib().trg_oracle_svd_u3(_f(A), _f(U))
        return U


def voxel_grid(xyz, leaf):
    """CPU restatement of the pcl::VoxelGrid step (trg_planner.cpp:91-94); parity unpinned (PCL)."""
    L = lib()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = np.empty_like(xyz)
    passthrough = C.c_int(0)
    L.trg_oracle_voxel_grid.restype = C.c_size_t
    L.trg_oracle_voxel_grid.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_float,
                                        C.POINTER(C.c_float), C.POINTER(C.c_int)]
    m = L.trg_oracle_voxel_grid(_f(xyz), xyz.shape[0], C.c_float(leaf), _f(out), C.byref(passthrough))
    return out[:m].copy(), bool(passthrough.value)


def algorithmic_bytes(n_points, counters, V, E):
    """B_alg = B_index + B_query + B_out."""
    b_index = (12 + 12 + 4) * n_points
    b_query = 12 * (counters["collision_hits"] + counters["ellipse_hits"] +
                    counters["nn_map_queries"])
    b_out = 16 * V + 4 * (V + 1) + 12 * E
    return dict(index=b_index, query=b_query, out=b_out, total=b_index + b_query + b_out)
