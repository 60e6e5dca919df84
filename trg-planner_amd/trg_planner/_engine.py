"""ctypes binding of csrc/libtrg_engine.so (C ABI: include/trg_engine.h).

The library is the product; this module adds nothing but argument marshalling.  There is no
fallback: if the shared library is missing, or no gfx950 device is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
# TRG_ENGINE_LIB: another build of the same library (A/B measurements of kernel variants on one box)
LIB_PATH = os.environ.get("TRG_ENGINE_LIB") or os.path.join(CSRC, "libtrg_engine.so")

KIND_GLOBAL, KIND_LOCAL, KIND_PRECLEAN, KIND_STITCHED = 0, 1, 2, 3
_KINDS = {"global": KIND_GLOBAL, "local": KIND_LOCAL, "preclean": KIND_PRECLEAN,
          "stitched": KIND_STITCHED}

STATUS_NAMES = {0: "TRG_OK", 1: "TRG_ERR_INVALID_ARG", 2: "TRG_ERR_NO_MAP", 3: "TRG_ERR_NO_ROOT",
                4: "TRG_ERR_DEVICE", 5: "TRG_ERR_NO_GRAPH", 6: "TRG_ERR_NOT_FOUND",
                7: "TRG_ERR_IO", 8: "TRG_ERR_CAPACITY"}


class TrgError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class TrgParams(C.Structure):
    _fields_ = [("is_verbose", C.c_int32), ("expand_dist", C.c_float), ("robot_size", C.c_float),
                ("sample_num", C.c_int32), ("height_threshold", C.c_float),
                ("collision_threshold", C.c_float), ("update_collision_threshold", C.c_float),
                ("safety_factor", C.c_float), ("goal_tolerance", C.c_float)]


class TrgSampler(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("table_bits", C.c_int32)]


class TrgCsrView(C.Structure):
    _fields_ = [("num_nodes", C.c_int32), ("num_edges", C.c_int32),
                ("node_xyz", C.POINTER(C.c_float)), ("node_state", C.POINTER(C.c_int32)),
                ("rowptr", C.POINTER(C.c_int32)), ("col", C.POINTER(C.c_int32)),
                ("weight", C.POINTER(C.c_float)), ("dist", C.POINTER(C.c_float)),
                ("creation_id", C.POINTER(C.c_int32))]


class TrgPathInfo(C.Structure):
    _fields_ = [("direct_dist", C.c_float), ("path_length", C.c_float), ("avg_risk", C.c_float),
                ("num_points", C.c_int32)]


class TrgStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "map_points", "expanded_nodes", "trials", "samples", "created_nodes", "invalid_nodes",
        "edge_calls", "edge_evals_gpu", "nn_ties", "gate_uncertain", "sync_batches",
        "bytes_sample_kernel", "bytes_spec_kernel", "bytes_edge_kernel", "bytes_index_build")] + [
        ("ms_index_build", C.c_double), ("ms_sample_kernel", C.c_double),
        ("ms_spec_kernel", C.c_double), ("ms_edge_kernel", C.c_double),
        ("launches_sample_kernel", C.c_uint64), ("launches_spec_kernel", C.c_uint64),
        ("launches_edge_kernel", C.c_uint64), ("ms_set_map_total", C.c_double),
        ("ms_init_graph_total", C.c_double), ("ms_replay_host", C.c_double),
        ("ms_finalize_host", C.c_double), ("ms_wait_gpu", C.c_double),
        ("bfs_levels", C.c_uint64), ("used_device_bfs", C.c_uint64), ("bfs_fallbacks", C.c_uint64),
        ("bfs_max_spin", C.c_uint64), ("bfs_host_levels", C.c_uint64),
        ("map_nn_ties", C.c_uint64),
        ("ms_bfs_loop", C.c_double), ("ms_deferred", C.c_double),
        ("map_nn_resolved", C.c_uint64), ("map_nn_unresolved", C.c_uint64),
        ("bfs_tie_fixups", C.c_uint64), ("bytes_spec_created", C.c_uint64),
        ("ms_rare_events", C.c_double), ("bfs_ticket_reruns", C.c_uint64),
        ("bfs_multipass_rows", C.c_uint64), ("ms_upload", C.c_double), ("presampled_nodes", C.c_uint64)]


# every symbol include/trg_engine.h declares (tests check that the library exports all of them)
EXPORTS = [
    "trg_engine_create", "trg_engine_destroy", "trg_engine_last_error", "trg_engine_device_arch",
    "trg_engine_set_global_map", "trg_engine_set_global_map_device", "trg_engine_set_local_map",
    "trg_engine_reset_map", "trg_engine_reset_graph", "trg_engine_init_graph",
    "trg_engine_update_graph", "trg_engine_export_csr", "trg_engine_save_json",
    "trg_engine_load_json", "trg_engine_plan", "trg_engine_refine_path",
    "trg_engine_is_collision_batch", "trg_engine_nearest_z_batch", "trg_engine_edge_risk_batch",
    "trg_engine_is_frontier_batch", "trg_engine_get_stats", "trg_engine_get_sampler_table",
    "trg_engine_debug_map_index", "trg_engine_set_option", "trg_engine_fallback_reason",
    "trg_engine_check_reached", "trg_engine_check_replan", "trg_engine_set_tile",
    "trg_engine_voxel_filter", "trg_engine_plan_batch",
    "trg_engine_stitch_boundary", "trg_engine_stitch_cross", "trg_engine_stitch_assemble",
    "trg_engine_graph_sizes",
    "trg_engine_comm_unique_id", "trg_engine_comm_init", "trg_engine_comm_adopt", "trg_engine_comm_destroy",
    "trg_engine_stitch_exchange",
]


def build_library(force=False):
    """Compile csrc/*.hip for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import glob
    # every source the build reads: a stale library must never pass for the tree's code
    srcs = []
    for pat in ("*.hip", "*.inc", "*.ipp", "*.h", "*.cpp", "*.sh"):
        srcs += glob.glob(os.path.join(CSRC, pat))
    srcs += glob.glob(os.path.normpath(os.path.join(CSRC, "..", "..", "include", "*")))
    stale = force or not os.path.exists(LIB_PATH) or not glob.glob(os.path.join(_HERE, "_trg_pybind*.so"))
    if not stale:
        t = os.path.getmtime(LIB_PATH)
        stale = any(os.path.exists(s) and os.path.getmtime(s) > t for s in srcs)
    if stale:
        subprocess.check_call(["bash", os.path.join(CSRC, "build.sh")])
    return LIB_PATH


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run trg-planner_amd/csrc/build.sh "
                          "(__graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.trg_engine_create.argtypes = [C.POINTER(TrgParams), C.c_int, C.POINTER(vp)]
    L.trg_engine_destroy.argtypes = [vp]
    L.trg_engine_destroy.restype = None
    L.trg_engine_last_error.argtypes = [vp]
    L.trg_engine_last_error.restype = C.c_char_p
    L.trg_engine_device_arch.argtypes = [vp]
    L.trg_engine_device_arch.restype = C.c_char_p
    L.trg_engine_set_global_map.argtypes = [vp, fp, C.c_size_t, C.c_size_t]
    L.trg_engine_set_global_map_device.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
    L.trg_engine_set_local_map.argtypes = [vp, fp, fp, C.c_size_t, C.c_size_t]
    L.trg_engine_reset_map.argtypes = [vp, C.c_int]
    L.trg_engine_reset_graph.argtypes = [vp, C.c_int]
    L.trg_engine_init_graph.argtypes = [vp, fp, C.POINTER(TrgSampler)]
    L.trg_engine_update_graph.argtypes = [vp]
    L.trg_engine_export_csr.argtypes = [vp, C.c_int, C.POINTER(TrgCsrView)]
    L.trg_engine_save_json.argtypes = [vp, C.c_char_p]
    L.trg_engine_load_json.argtypes = [vp, C.c_char_p]
    L.trg_engine_plan.argtypes = [vp, fp, fp, fp, C.c_int32, C.POINTER(TrgPathInfo)]
    L.trg_engine_refine_path.argtypes = [fp, C.c_int32, fp, C.c_int32]
    L.trg_engine_refine_path.restype = C.c_int32
    L.trg_engine_plan_batch.argtypes = [vp, fp, fp, C.c_size_t, fp, C.c_int32,
                                        C.POINTER(C.c_int32), C.POINTER(TrgPathInfo)]
    L.trg_engine_voxel_filter.argtypes = [vp, fp, C.c_size_t, C.c_size_t, C.c_float, fp,
                                          C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]
    L.trg_engine_is_collision_batch.argtypes = [vp, C.c_int, C.c_float, fp, C.c_size_t, ip, ip, ip]
    L.trg_engine_nearest_z_batch.argtypes = [vp, C.c_int, fp, C.c_size_t, fp]
    L.trg_engine_edge_risk_batch.argtypes = [vp, C.c_int, fp, fp, C.c_size_t, ip, ip, fp, fp]
    L.trg_engine_is_frontier_batch.argtypes = [vp, fp, C.c_size_t, ip]
    L.trg_engine_get_stats.argtypes = [vp, C.POINTER(TrgStats)]
    L.trg_engine_get_sampler_table.argtypes = [vp, fp, fp]
    L.trg_engine_debug_map_index.argtypes = [vp, C.c_int, fp, fp, fp, ip, ip, fp]
    L.trg_engine_check_reached.argtypes = [vp, fp]
    L.trg_engine_check_reached.restype = C.c_int32
    L.trg_engine_check_replan.argtypes = [vp, fp, fp, C.c_int32]
    L.trg_engine_check_replan.restype = C.c_int32
    L.trg_engine_set_tile.argtypes = [vp, fp, C.c_uint32]
    L.trg_engine_stitch_boundary.argtypes = [vp, fp, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, ip]
    L.trg_engine_stitch_cross.argtypes = [vp, C.c_int32, C.c_int32, vp, ip, vp, C.c_int32, ip]
    L.trg_engine_stitch_assemble.argtypes = [vp, C.c_int32, C.c_int32, ip, vp, C.c_int32]
    L.trg_engine_graph_sizes.argtypes = [vp, C.c_int, ip, ip]
    u8p = C.POINTER(C.c_uint8)
    L.trg_engine_comm_unique_id.argtypes = [vp, u8p]
    L.trg_engine_comm_init.argtypes = [vp, u8p, C.c_int32, C.c_int32]
    L.trg_engine_comm_adopt.argtypes = [vp, vp]
    L.trg_engine_comm_destroy.argtypes = [vp]
    L.trg_engine_stitch_exchange.argtypes = [vp, fp, C.c_int32, C.c_int32, ip, ip]
    L.trg_engine_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.trg_engine_fallback_reason.argtypes = [vp]
    L.trg_engine_fallback_reason.restype = C.c_char_p
    _lib = L
    return L


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class CsrGraph:
    """Host copy of a TrgCsrView."""

    def __init__(self, xyz, state, rowptr, col, weight, dist, cid):
        self.xyz, self.state, self.rowptr, self.col = xyz, state, rowptr, col
        self.w, self.dist, self.cid = weight, dist, cid

    @property
    def V(self):
        return int(self.state.shape[0])

    @property
    def E(self):
        return int(self.col.shape[0])


class Engine:
    """One TrgEngine handle."""

    def __init__(self, expand_dist=0.6, robot_size=0.3, sample_num=7, height_threshold=0.16,
                 collision_threshold=0.1, update_collision_threshold=0.5, safety_factor=3.0,
                 goal_tolerance=0.8, is_verbose=False, device=0):
        self.L = load_library()
        self.params = TrgParams(int(is_verbose), expand_dist, robot_size, sample_num,
                                height_threshold, collision_threshold, update_collision_threshold,
                                safety_factor, goal_tolerance)
        self.h = C.c_void_p()
        st = self.L.trg_engine_create(C.byref(self.params), device, C.byref(self.h))
        if st != 0:
            msg = self.L.trg_engine_last_error(self.h).decode() if self.h else "create failed"
            if self.h:
                self.L.trg_engine_destroy(self.h)
                self.h = C.c_void_p()
            raise TrgError(st, msg)
        self.sampler = TrgSampler(1, 16)

    def close(self):
        if getattr(self, "h", None):
            self.L.trg_engine_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        if st != 0:
            raise TrgError(st, self.L.trg_engine_last_error(self.h).decode())

    @property
    def arch(self):
        return self.L.trg_engine_device_arch(self.h).decode()

    def set_tile(self, core_xyxy=None, epoch=0):
        """Tiled builds: node creation restricted to [x0,x1) x [y0,y1); sampler epoch of the tile."""
        if core_xyxy is None:
            self._chk(self.L.trg_engine_set_tile(self.h, None, int(epoch)))
        else:
            c = np.ascontiguousarray(core_xyxy, dtype=np.float32)
            self._chk(self.L.trg_engine_set_tile(self.h, _f(c), int(epoch)))

    # ---- tiled builds: boundary stitch (device buffers = torch tensors' data_ptr()) -------------------
    def stitch_boundary(self, core_xyxy, cols, rows, tile, rec_ptr=None, cap=0):
        """Number of boundary nodes; with rec_ptr (device, cap x 16 bytes) also their records."""
        c = np.ascontiguousarray(core_xyxy, dtype=np.float32)
        n = C.c_int32(0)
        self._chk(self.L.trg_engine_stitch_boundary(self.h, _f(c), cols, rows, tile,
                                                    C.c_void_p(rec_ptr or 0), cap, C.byref(n)))
        return n.value

    def stitch_cross(self, tile, ntiles, all_rec_ptr, rec_offsets, edges_ptr=None, cap=0):
        off = np.ascontiguousarray(rec_offsets, dtype=np.int32)
        n = C.c_int32(0)
        self._chk(self.L.trg_engine_stitch_cross(self.h, tile, ntiles, C.c_void_p(all_rec_ptr or 0), _i(off),
                                                 C.c_void_p(edges_ptr or 0), cap, C.byref(n)))
        return n.value

    def stitch_assemble(self, tile, ntiles, node_offsets, all_edges_ptr, n_edges):
        off = np.ascontiguousarray(node_offsets, dtype=np.int32)
        self._chk(self.L.trg_engine_stitch_assemble(self.h, tile, ntiles, _i(off),
                                                    C.c_void_p(all_edges_ptr or 0), n_edges))

    # ---- the native exchange: one call = the whole stitch of this rank's tile over RCCL -------------------
    def comm_unique_id(self):
        """128 bytes one rank draws and hands to every rank (e.g. torch.distributed broadcast)."""
        buf = (C.c_uint8 * 128)()
        self._chk(self.L.trg_engine_comm_unique_id(self.h, buf))
        return bytes(buf)

    def comm_init(self, unique_id, nranks, rank):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._chk(self.L.trg_engine_comm_init(self.h, buf, nranks, rank))
        self._comm_ranks = nranks

    def comm_destroy(self):
        self.L.trg_engine_comm_destroy(self.h)
        self._comm_ranks = 0

    def stitch_exchange(self, core_xyxy, cols, rows):
        """(boundary records of all tiles, cross edges of all tiles); rows: graph('stitched')."""
        c = np.ascontiguousarray(core_xyxy, dtype=np.float32)
        nb, nc = C.c_int32(0), C.c_int32(0)
        self._chk(self.L.trg_engine_stitch_exchange(self.h, _f(c), cols, rows, C.byref(nb), C.byref(nc)))
        return nb.value, nc.value

    def set_option(self, key, value):
        self._chk(self.L.trg_engine_set_option(self.h, str(key).encode(), str(value).encode()))

    @property
    def fallback_reason(self):
        return self.L.trg_engine_fallback_reason(self.h).decode()

    def set_sampler(self, seed=1, table_bits=16):
        self.sampler = TrgSampler(seed, table_bits)

    def set_global_map(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        self._chk(self.L.trg_engine_set_global_map(self.h, _f(xyz), xyz.shape[0], xyz.shape[1]))

    def set_global_map_device(self, data_ptr, n, stride=3):
        """Points already in HBM (e.g. a torch tensor's data_ptr())."""
        self._chk(self.L.trg_engine_set_global_map_device(self.h, C.c_void_p(data_ptr), n, stride))

    def set_local_map(self, start2d, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        s = np.ascontiguousarray(start2d, dtype=np.float32)
        self._chk(self.L.trg_engine_set_local_map(self.h, _f(s), _f(xyz), xyz.shape[0], 3))

    def reset_map(self, kind="global"):
        self._chk(self.L.trg_engine_reset_map(self.h, _KINDS[kind]))

    def reset_graph(self, kind="global"):
        self._chk(self.L.trg_engine_reset_graph(self.h, _KINDS[kind]))

    def init_graph(self, start3d):
        s = np.ascontiguousarray(start3d, dtype=np.float32)
        self._chk(self.L.trg_engine_init_graph(self.h, _f(s), C.byref(self.sampler)))

    def update_graph(self):
        self._chk(self.L.trg_engine_update_graph(self.h))

    def graph(self, kind="global"):
        v = TrgCsrView()
        self._chk(self.L.trg_engine_export_csr(self.h, _KINDS[kind], C.byref(v)))
        V, E = v.num_nodes, v.num_edges

        def arr(ptr, n, dt):
            if n == 0:
                return np.empty(0, dt)
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)

        return CsrGraph(arr(v.node_xyz, 3 * V, np.float32).reshape(V, 3),
                        arr(v.node_state, V, np.int32), arr(v.rowptr, V + 1, np.int32),
                        arr(v.col, E, np.int32), arr(v.weight, E, np.float32),
                        arr(v.dist, E, np.float32), arr(v.creation_id, V, np.int32))

    def graph_sizes(self, kind="global"):
        """(num_nodes, num_edges) without copying the arrays."""
        V, E = C.c_int32(0), C.c_int32(0)
        self._chk(self.L.trg_engine_graph_sizes(self.h, _KINDS[kind], C.byref(V), C.byref(E)))
        return int(V.value), int(E.value)

    def node_xyz(self, kind="global"):
        """(V, 3) float32 node positions only (no edge arrays copied)."""
        v = TrgCsrView()
        self._chk(self.L.trg_engine_export_csr(self.h, _KINDS[kind], C.byref(v)))
        if v.num_nodes == 0:
            return np.zeros((0, 3), np.float32)
        return np.ctypeslib.as_array(v.node_xyz, shape=(3 * v.num_nodes,)).astype(
            np.float32, copy=True).reshape(v.num_nodes, 3)

    def save_json(self, path):
        self._chk(self.L.trg_engine_save_json(self.h, str(path).encode()))

    def load_json(self, path):
        self._chk(self.L.trg_engine_load_json(self.h, str(path).encode()))

    def plan(self, start2d, goal3d, max_points=100000):
        s = np.ascontiguousarray(start2d, dtype=np.float32)
        g = np.ascontiguousarray(goal3d, dtype=np.float32)
        path = np.empty((max_points, 3), np.float32)
        info = TrgPathInfo()
        st = self.L.trg_engine_plan(self.h, _f(s), _f(g), _f(path), max_points, C.byref(info))
        if st == 6:  # TRG_ERR_NOT_FOUND: planSafePath returned false
            return np.empty((0, 3), np.float32), info
        self._chk(st)
        return path[:info.num_points].copy(), info

    def plan_batch(self, starts2d, goals3d, path_cap=200000):
        """m consecutive planSafePath calls in one boundary crossing -> list of (path, info)."""
        s = np.ascontiguousarray(starts2d, dtype=np.float32).reshape(-1, 2)
        g = np.ascontiguousarray(goals3d, dtype=np.float32).reshape(-1, 3)
        m = s.shape[0]
        path = np.empty((path_cap, 3), np.float32)
        off = np.zeros(m + 1, np.int32)
        infos = (TrgPathInfo * max(m, 1))()
        self._chk(self.L.trg_engine_plan_batch(self.h, _f(s), _f(g), m, _f(path), path_cap,
                                               off.ctypes.data_as(C.POINTER(C.c_int32)), infos))
        return [(path[off[k]:off[k + 1]].copy(), infos[k]) for k in range(m)]

    def check_reached(self, pos2d):
        p = np.ascontiguousarray(pos2d, dtype=np.float32)
        return bool(self.L.trg_engine_check_reached(self.h, _f(p)))

    def check_replan(self, pos2d, path):
        p = np.ascontiguousarray(pos2d, dtype=np.float32)
        path = np.ascontiguousarray(path, dtype=np.float32).reshape(-1, 3)
        return bool(self.L.trg_engine_check_replan(self.h, _f(p), _f(path), path.shape[0]))

    def refine_path(self, path):
        path = np.ascontiguousarray(path, dtype=np.float32).reshape(-1, 3)
        out = np.empty((2 * path.shape[0] + 2, 3), np.float32)
        n = self.L.trg_engine_refine_path(_f(path), path.shape[0], _f(out), out.shape[0])
        return out[:n].copy()

    def is_collision(self, xy, kind="global", threshold=None):
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        m = xy.shape[0]
        if threshold is None:
            threshold = self.params.collision_threshold
        flag = np.empty(m, np.int32)
        cnt = np.empty(m, np.int32)
        n = np.empty(m, np.int32)
        self._chk(self.L.trg_engine_is_collision_batch(self.h, _KINDS[kind], threshold, _f(xy), m,
                                                       _i(flag), _i(cnt), _i(n)))
        return flag, cnt, n

    def nearest_z(self, xy, kind="global"):
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        z = np.empty(xy.shape[0], np.float32)
        self._chk(self.L.trg_engine_nearest_z_batch(self.h, _KINDS[kind], _f(xy), xy.shape[0], _f(z)))
        return z

    def edge_risk(self, p1, p2, kind="global"):
        p1 = np.ascontiguousarray(p1, dtype=np.float32).reshape(-1, 3)
        p2 = np.ascontiguousarray(p2, dtype=np.float32).reshape(-1, 3)
        m = p1.shape[0]
        status = np.empty(m, np.int32)
        n_pts = np.empty(m, np.int32)
        w = np.empty(m, np.float32)
        d = np.empty(m, np.float32)
        self._chk(self.L.trg_engine_edge_risk_batch(self.h, _KINDS[kind], _f(p1), _f(p2), m,
                                                    _i(status), _i(n_pts), _f(w), _f(d)))
        return status, n_pts, w, d

    def voxel_filter(self, xyz, leaf):
        """pcl::VoxelGrid of TRGPlanner::loadPrebuiltMap (trg_planner.cpp:91-94) on the GPU."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        out = np.empty_like(xyz)
        n_out = C.c_size_t(0)
        passthrough = C.c_int32(0)
        self._chk(self.L.trg_engine_voxel_filter(self.h, _f(xyz), xyz.shape[0], 3, C.c_float(leaf),
                                                 _f(out), C.byref(n_out), C.byref(passthrough)))
        return out[:n_out.value].copy()

    def is_frontier(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        flag = np.empty(xy.shape[0], np.int32)
        self._chk(self.L.trg_engine_is_frontier_batch(self.h, _f(xy), xy.shape[0], _i(flag)))
        return flag

    def stats(self):
        s = TrgStats()
        self._chk(self.L.trg_engine_get_stats(self.h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in TrgStats._fields_}

    def sampler_table(self):
        n = 1 << self.sampler.table_bits
        c = np.empty(n, np.float32)
        s = np.empty(n, np.float32)
        self._chk(self.L.trg_engine_get_sampler_table(self.h, _f(c), _f(s)))
        return c, s

    def map_index(self, kind="global", n=None):
        wh = np.zeros(2, np.int32)
        org = np.zeros(3, np.float32)
        self._chk(self.L.trg_engine_debug_map_index(self.h, _KINDS[kind], None, None, None, None,
                                                    _i(wh), _f(org)))
        if n is None:
            return wh, org
        x = np.empty(n, np.float32)
        y = np.empty(n, np.float32)
        z = np.empty(n, np.float32)
        p = np.empty(n, np.int32)
        self._chk(self.L.trg_engine_debug_map_index(self.h, _KINDS[kind], _f(x), _f(y), _f(z),
                                                    _i(p), _i(wh), _f(org)))
        return x, y, z, p, wh, org
