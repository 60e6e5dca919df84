/*
 * include/trg_engine.h -- C ABI of the MI355X-native Traversal-Risk-Graph construction engine.
 *
 * This is the drop-in boundary for the one hot path this repository accelerates: everything the
 * reference's `class TRG` does between "here is a point-cloud map" and "here is the graph"
 * (reference: cpp/trg_planner/core/trg_planner/include/graph/trg.h:50-98 and
 * src/graph/trg.cpp).  Plain pointers and sizes only; no C++/torch types; no exceptions cross
 * the boundary; every call returns a TrgStatus and trg_engine_last_error() explains failures.
 *
 * Each entry point cites the reference interface it replaces.  The C++ `TRG`/`TRGPlanner`
 * shims, the Python mirror (trg-planner_amd/trg_planner) and the reference-side binding shown
 * in INTEGRATION.md are thin layers over exactly these symbols.
 *
 * Threading: like the reference (every entry is taken under TRG::mtx.graph, trg.cpp:37,196,458,610)
 * an engine must be entered from one thread at a time; one HIP stream set per engine.
 */
#ifndef TRG_ENGINE_H_
#define TRG_ENGINE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct TrgEngine TrgEngine;

typedef enum TrgStatus {
  TRG_OK = 0,
  TRG_ERR_INVALID_ARG = 1,
  TRG_ERR_NO_MAP = 2,        /* reference: assert "Map is empty", trg.cpp:42 */
  TRG_ERR_NO_ROOT = 3,       /* reference: "Failed to generate root node" + exit(1), trg.cpp:49-52 */
  TRG_ERR_DEVICE = 4,        /* HIP runtime error, no GPU, wrong architecture */
  TRG_ERR_NO_GRAPH = 5,
  TRG_ERR_NOT_FOUND = 6,     /* planSafePath returned false, trg.cpp:689 */
  TRG_ERR_IO = 7,
  TRG_ERR_CAPACITY = 8
} TrgStatus;

/* Graph / map selector: reference trgMap_ keys "global" / "local" (trg.h:113-119).
 * TRG_KIND_PRECLEAN is build-side instrumentation: the global graph as it stood right before
 * the last cleanGraph(), ids == creation order (used by the parity tests). */
typedef enum TrgKind {
  TRG_KIND_GLOBAL = 0,
  TRG_KIND_LOCAL = 1,
  TRG_KIND_PRECLEAN = 2,
  TRG_KIND_STITCHED = 3 /* tiled builds: this tile's rows of the stitched global graph (global ids) */
} TrgKind;

/* Node states: reference TRG::NodeState, trg.h:27-31 */
enum { TRG_NODE_VALID = 0, TRG_NODE_INVALID = -1, TRG_NODE_FRONTIER = 1 };

/* Constructor arguments of reference TRG::TRG, trg.h:51-59 / trg.cpp:11-34 (same order). */
typedef struct TrgParams {
  int32_t is_verbose;
  float expand_dist;
  float robot_size;
  int32_t sample_num;
  float height_threshold;
  float collision_threshold;
  float update_collision_threshold;
  float safety_factor;
  float goal_tolerance;
} TrgParams;

/* The reference seeds std::mt19937 from std::random_device (trg.cpp:20), so it has no canonical
 * sample stream; here the sampler is an explicit input.  Direction of trial t of the expansion
 * of node id in build epoch e = table[hash(seed, e, id, t) >> (32 - table_bits)], table[k] =
 * (cosf, sinf)((float)(k / 2^bits * 2 * M_PI)) computed once with the host libm (same values the
 * reference's `distr_(gen_) * 2 * M_PI` -> cos/sin would give for that draw, trg.cpp:395-397). */
typedef struct TrgSampler {
  uint32_t seed;
  int32_t table_bits; /* 2..20 (out of range: 16); few bits = few directions: degenerate, tie-rich graphs for tests */
} TrgSampler;

/* Read-only view of a built graph in CSR form (host memory owned by the engine, valid until the
 * next call that mutates that graph).  Replaces the std::unordered_map<int, Node*> the reference
 * hands out (TRG::getGraph / getGraphCopy, trg.cpp:805-824): row i is the node with id_ == i,
 * col/weight/dist are that node's edges_ in order (Edge{dst_id_, weight_, dist_}, trg.h:20-25). */
typedef struct TrgCsrView {
  int32_t num_nodes;
  int32_t num_edges;        /* directed edges = CSR nnz */
  const float *node_xyz;    /* num_nodes x 3, Node::pos_ */
  const int32_t *node_state;/* num_nodes, Node::state_ */
  const int32_t *rowptr;    /* num_nodes + 1 */
  const int32_t *col;       /* num_edges, Edge::dst_id_ */
  const float *weight;      /* num_edges, Edge::weight_ */
  const float *dist;        /* num_edges, Edge::dist_ */
  const int32_t *creation_id; /* num_nodes: index of the node in creation order of the last build */
} TrgCsrView;

/* Output of trg_engine_plan: reference TRG::planSafePath out-params (trg.cpp:603-608). */
typedef struct TrgPathInfo {
  float direct_dist;
  float path_length;
  float avg_risk;
  int32_t num_points;
} TrgPathInfo;

/* Counters and timers of the last build (instrumentation; bench.py's roofline uses them). */
typedef struct TrgStats {
  uint64_t map_points;
  uint64_t expanded_nodes;     /* nodes popped from the BFS queue */
  uint64_t trials;             /* sample draws */
  uint64_t samples;            /* accepted samples */
  uint64_t created_nodes;
  uint64_t invalid_nodes;
  uint64_t edge_calls;         /* wireEdge() calls replayed */
  uint64_t edge_evals_gpu;     /* edge evaluations executed on the GPU (speculative ones included) */
  uint64_t nn_ties;            /* exact fp32 distance ties between nearest-NODE candidates; resolved
                                  exactly (reference kd-tree traversal order, host_index.h) */
  uint64_t gate_uncertain;     /* slope gates decided by host libm atan2f */
  uint64_t sync_batches;       /* synchronous GPU round trips forced by the replay */
  /* bytes of map points inside query radii that the GPU kernels touched (12 B per hit) */
  uint64_t bytes_sample_kernel;   /* sampling discs: k_level_sample (device BFS) / k_sample_nodes */
  uint64_t bytes_spec_kernel;     /* speculative parent edges: k_level_spec / k_spec_edges */
  uint64_t bytes_edge_kernel;     /* deferred wireEdge evaluations: k_calls_gather / k_edges */
  uint64_t bytes_index_build;
  /* device time per kernel, milliseconds, measured with hipEvents on the launch stream.  Inside the
   * device-resident BFS k_level_sample + k_level_spec are timed TOGETHER on every 8th level only (an
   * event pair costs ~12 us of stream time per level), reported as ms_sample_kernel /
   * launches_sample_kernel (ms_spec_kernel stays 0) and scaled by launches / timed launches; the
   * deferred edge evaluations are timed in full. */
  double ms_index_build;
  double ms_sample_kernel;
  double ms_spec_kernel;
  double ms_edge_kernel;
  uint64_t launches_sample_kernel;
  uint64_t launches_spec_kernel;
  uint64_t launches_edge_kernel;
  /* host wall time, milliseconds */
  double ms_set_map_total;
  double ms_init_graph_total;
  double ms_replay_host;
  double ms_finalize_host;
  double ms_wait_gpu;
  uint64_t bfs_levels;         /* BFS depth of the last build (device-resident path) */
  uint64_t used_device_bfs;    /* 1: BFS + CSR ran on the GPU; 0: host replay */
  uint64_t bfs_fallbacks;      /* device path declined and the host replay redid the build */
  uint64_t bfs_max_spin;       /* longest dependency wait (poll iterations) in k_level_resolve */
  uint64_t bfs_host_levels;    /* BFS levels replayed on the host because of an exact distance tie */
  uint64_t map_nn_ties;        /* nearest-map-point queries (trial discs and elevation lookups,
                                  trg.cpp:244-247) that met two MAP points at exactly the same fp32
                                  distance (~2e-7 per query) */
  double ms_bfs_loop;          /* device path: wall time of the level loop */
  double ms_deferred;          /* device path: wall time of the deferred edge evaluations */
  uint64_t map_nn_resolved;    /* of those, the ones an accepted sample / addNode depended on: decided
                                  in the reference's map-tree visiting order (kdtree.c:303-362) */
  uint64_t map_nn_unresolved;  /* ties left at "lowest cloud index" (more than 16 points tied, or more
                                  than 256 tied samples in one launch) -- 0 in practice */
  uint64_t bfs_tie_fixups;     /* BFS levels whose only trouble was a distance tie among nodes that existed
                                  before the level: the reference's winner was handed to the device and
                                  resolve + commit ran again (no host replay) */
  uint64_t bytes_spec_created; /* device path: the part of bytes_spec_kernel spent on the parent edges of
                                  the nodes that were created -- the wireEdge(node, new_node) calls the
                                  reference itself evaluates (trg.cpp:425); the rest of
                                  bytes_spec_kernel is speculation on candidates that merged */
  double ms_rare_events;       /* device path: wall time inside the level loop spent repairing rare events
                                  (map-point ties, uncertain slope gates, node ties, host level replays) */
  uint64_t bfs_ticket_reruns;  /* device path: resolve launches repeated with start tickets as workgroup indices
                                  after a bounded inter-workgroup wait ran out */
  uint64_t bfs_multipass_rows; /* device path: sample slots whose blockers did not fit one row (taken in passes) */
  double ms_upload;            /* host cloud -> HBM of the last setGlobalMap / setLocalMap from a host pointer */
  uint64_t presampled_nodes;   /* device path: expanded nodes whose samples were already there when their level's
                                  sampling kernel started (drawn inside the previous level's resolve launch) */
} TrgStats;

/* ---- lifetime ------------------------------------------------------------------------------- */
/* reference: TRG::TRG(...) trg.cpp:11-34.  device = HIP device ordinal. */
TrgStatus trg_engine_create(const TrgParams *params, int device, TrgEngine **out);
void trg_engine_destroy(TrgEngine *e);
const char *trg_engine_last_error(const TrgEngine *e);
/* "gfx950" etc. of the device the engine runs on */
const char *trg_engine_device_arch(const TrgEngine *e);

/* ---- map ingest ------------------------------------------------------------------------------ */
/* reference: TRG::setGlobalMap(PointCloudPtr&) trg.cpp:179-193 (the kd_insert2 loop becomes the
 * cell-sorted SoA index build on the GPU).  xyz: n points, `stride` floats apart (3 for packed
 * xyz, 4 for pcl::PointXYZ). */
TrgStatus trg_engine_set_global_map(TrgEngine *e, const float *xyz, size_t n, size_t stride);
/* same, points already resident in device memory (HBM) */
TrgStatus trg_engine_set_global_map_device(TrgEngine *e, const float *d_xyz, size_t n, size_t stride);
/* reference: TRG::setLocalMap(Vector2f start2d, PointCloudPtr&) trg.cpp:195-209 (also refreshes
 * the local graph membership, trg.cpp:211-231) */
TrgStatus trg_engine_set_local_map(TrgEngine *e, const float start_xy[2], const float *xyz, size_t n,
                                   size_t stride);
/* reference: TRG::resetMap / TRG::resetGraph, trg.cpp:732-744 */
TrgStatus trg_engine_reset_map(TrgEngine *e, TrgKind kind);
TrgStatus trg_engine_reset_graph(TrgEngine *e, TrgKind kind);

/* ---- graph build ----------------------------------------------------------------------------- */
/* reference: TRG::initGraph(bool isPreMap, Vector3f start3d) trg.cpp:36-64
 * (root seeding -> expandGraph :372-454 -> cleanGraph(false) :491-535). */
TrgStatus trg_engine_init_graph(TrgEngine *e, const float start_xyz[3], const TrgSampler *sampler);
/* reference: TRG::updateGraph() trg.cpp:456-489 */
TrgStatus trg_engine_update_graph(TrgEngine *e);

/* ---- export ---------------------------------------------------------------------------------- */
/* reference: TRG::getGraph / getGraphCopy trg.cpp:805-824 */
TrgStatus trg_engine_export_csr(TrgEngine *e, TrgKind kind, TrgCsrView *out);
/* node / directed-edge counts of a graph without touching (or, for TRG_KIND_STITCHED, fetching) its arrays */
TrgStatus trg_engine_graph_sizes(TrgEngine *e, TrgKind kind, int32_t *num_nodes, int32_t *num_edges);
/* reference: TRG::saveGraph / loadPrebuiltGraph trg.cpp:66-177 (same JSON schema) */
TrgStatus trg_engine_save_json(TrgEngine *e, const char *path);
TrgStatus trg_engine_load_json(TrgEngine *e, const char *path);

/* ---- query (host A*, consumes the CSR) ------------------------------------------------------- */
/* reference: TRG::planSafePath trg.cpp:603-690 (+ setGoal :537-565).  path_xyz receives up to
 * max_points x 3 floats; info->num_points is the full length. */
TrgStatus trg_engine_plan(TrgEngine *e, const float start_xy[2], const float goal_xyz[3],
                          float *path_xyz, int32_t max_points, TrgPathInfo *info);
/* m consecutive planSafePath calls in one boundary crossing (reference: the loop over start/goal
 * pairs of python/examples/run_trg_planner.py:35-43).  Exactly the results
 * of calling trg_engine_plan m times in order (the goal state left behind is that of the last
 * query).  path_xyz: room for path_cap points in total; query k's points are
 * [offsets[k], offsets[k+1]) (offsets has m + 1 entries); a query without a path has an empty
 * range and infos[k].num_points == 0; a path that no longer fits is truncated (num_points keeps
 * its full length). */
TrgStatus trg_engine_plan_batch(TrgEngine *e, const float *starts_xy, const float *goals_xyz,
                                size_t m, float *path_xyz, int32_t path_cap, int32_t *offsets,
                                TrgPathInfo *infos);
/* reference: TRG::checkReadched (sic) trg.cpp:567-574 / TRG::checkReplan trg.cpp:576-601; 1 = true */
int32_t trg_engine_check_reached(TrgEngine *e, const float pos_xy[2]);
int32_t trg_engine_check_replan(TrgEngine *e, const float pos_xy[2], const float *path_xyz,
                                int32_t n_path);
/* reference: TRG::refinePath trg.cpp:692-730.  Returns the number of output points. */
int32_t trg_engine_refine_path(const float *in_xyz, int32_t n_in, float *out_xyz, int32_t max_out);

/* ---- batched probes of the pure map functions (debug / parity tests) ------------------------- */
/* reference: TRG::isCollision(pos, type, threshold) trg.cpp:746-778.  Any output may be NULL.
 * flag: 1 = collision; cnt: points with |z - z_med| > height_threshold; n: points in the disc. */
TrgStatus trg_engine_is_collision_batch(TrgEngine *e, TrgKind map, float threshold, const float *xy,
                                        size_t m, int32_t *flag, int32_t *cnt, int32_t *n);
/* reference: the kd_nearest2 elevation lookup of TRG::addNode trg.cpp:244-247 */
TrgStatus trg_engine_nearest_z_batch(TrgEngine *e, TrgKind map, const float *xy, size_t m, float *z);
/* reference: the position-only part of TRG::wireEdge trg.cpp:269-363.
 * status: 0 ok, 1 slope gate, 2 segment collision, 3 empty gather, 4 fewer than 3 points. */
TrgStatus trg_engine_edge_risk_batch(TrgEngine *e, TrgKind map, const float *p1_xyz,
                                     const float *p2_xyz, size_t m, int32_t *status, int32_t *n_pts,
                                     float *weight, float *dist);
/* reference: TRG::isFrontier trg.cpp:780-803 */
TrgStatus trg_engine_is_frontier_batch(TrgEngine *e, const float *xy, size_t m, int32_t *flag);

/* ---- map ingest: voxel-grid filter --------------------------------------------------------------- */
/* reference: the pcl::VoxelGrid step of TRGPlanner::loadPrebuiltMap, trg_planner.cpp:91-94
 * (setLeafSize(voxelSize x3), filter): one point per occupied voxel = centroid of its points,
 * voxels in ascending voxel index (PCL filters/impl/voxel_grid.hpp applyFilter; PCL is not vendored
 * by the reference -> parity unpinned, the in-voxel fp32 summation order is ascending point index).
 * xyz: n host points with `stride` floats each; out_xyz: room for 3*n floats; *n_out = points
 * written.  *passthrough (optional) = 1 when the leaf is too small for 32-bit voxel indices and the
 * input was handed through unchanged, as PCL does. */
TrgStatus trg_engine_voxel_filter(TrgEngine *e, const float *xyz, size_t n, size_t stride, float leaf,
                                  float *out_xyz, size_t *n_out, int32_t *passthrough);

/* ---- options ----------------------------------------------------------------------------------- */
/* "replay" = "device" (default: BFS, dedupe and CSR on the GPU when expandGraph's step 3 is off,
 * trg.cpp:429) | "host" (sequential replay on the host, the only mode for step-3 configs);
 * "keep_preclean" = "0" | "1" (keep the TRG_KIND_PRECLEAN snapshot); "defer_overlap" = "1" | "0" | "2"
 * (device BFS: deferred wireEdge evaluations pipelined behind the level loop on a second stream, one
 * batch per level -- default; 0: after the loop; 2: only pair-table inserts and selection beside the loop);
 * "tie_inplace" = "1" | "0" (device BFS: a nearest-node distance tie is settled for the affected slot alone on
 * the committed level -- off: the whole level is replayed on the host).  All modes give identical
 * graphs; the env var TRG_REPLAY=host sets the default.  Test hooks (never change results):
 * "debug_tie_every" = n (treat every n-th BFS level as tie-affected -> host level replay),
 * "debug_gate_margin" = x (widen the band of slope gates left to the host's libm),
 * "debug_spec_bound" = n (cap the speculative next-level sampling launch at n nodes -> top-up
 * launches), "debug_fallback_level" = n (the device BFS declines at level n -> whole-build host
 * replay), "debug_stall_level" = n (k_level_resolve leaves one candidate of level n undecided ->
 * BFS_ERR_STALL -> the level is taken back and replayed on the host), "debug_lookback_level" = n (one
 * workgroup's commit look-back gives up at level n -> BFS_ERR_LOOKBACK -> whole-build host replay). */
TrgStatus trg_engine_set_option(TrgEngine *e, const char *key, const char *value);
/* Tiled builds (multi-GPU; an extension, not a reference interface): restrict
 * node creation to the core region [x0,x1) x [y0,y1) -- a sample outside it counts as a rejected
 * draw -- and give the tile its own sampler epoch.  core_xyxy == NULL restores the whole plane. */
TrgStatus trg_engine_set_tile(TrgEngine *e, const float core_xyxy[4], uint32_t epoch);
/* ---- tiled builds: the boundary stitch (an extension; rule: trg_planner/tiled.py) --
 * Three steps with one exchange between each (all-gather-v of DEVICE buffers over RCCL, done by the
 * caller; tile index == rank).  d_* arguments are device pointers owned by the caller. */
typedef struct TrgBoundaryRec { int32_t local_id; float x, y, z; } TrgBoundaryRec;          /* 16 bytes */
typedef struct TrgCrossEdge { int32_t tile_a, id_a, tile_b, id_b; float weight, dist; } TrgCrossEdge; /* 24 */
/* 1: the nodes of this tile closer than expand_dist to a core side shared with another tile of the
 * cols x rows grid, ascending local id.  d_rec may be NULL to ask for the count only. */
TrgStatus trg_engine_stitch_boundary(TrgEngine *e, const float core_xyxy[4], int32_t cols, int32_t rows,
                                     int32_t tile, TrgBoundaryRec *d_rec, int32_t cap, int32_t *n_out);
/* 2: the cross edges this tile owns: every pair (a of this tile, b of a HIGHER tile) of the gathered
 * records (all tiles concatenated in tile order, rec_offsets[ntiles + 1] on the host) with fp32 planar
 * distance < expand_dist (the test of trg.cpp:414) whose wireEdge position-only part (trg.cpp:269-363)
 * succeeds on this tile's map; order (other tile, a, b). */
TrgStatus trg_engine_stitch_cross(TrgEngine *e, int32_t tile, int32_t ntiles, const TrgBoundaryRec *d_all_rec,
                                  const int32_t *rec_offsets, TrgCrossEdge *d_edges, int32_t cap,
                                  int32_t *n_out);
/* 3: this tile's rows of the global graph from ALL tiles' cross edges (any order): local edges with
 * global ids (node_offsets[tile] + local id; node_offsets[ntiles + 1] = exclusive prefix of the tiles'
 * node counts), then the row's cross edges ordered by the global id of their other end.  Read the
 * result with trg_engine_export_csr(TRG_KIND_STITCHED); creation_id holds the rows' global ids. */
TrgStatus trg_engine_stitch_assemble(TrgEngine *e, int32_t tile, int32_t ntiles, const int32_t *node_offsets,
                                     const TrgCrossEdge *d_all_edges, int32_t n_edges);
/* The whole stitch of this rank's tile with both exchanges done natively over RCCL (librccl is resolved at
 * run time): steps 1-3 above with an all-gather of counts + an all-gather of padded payloads between them.
 * rank = tile, ranks = cols * rows.  The communicator is either made here from a unique id that ONE rank
 * draws and the application hands to every rank (MPI_Bcast, a file, torch.distributed ...), or adopted
 * from the caller (an ncclComm_t of the SAME librccl instance).  A rank whose own step fails announces it
 * in the next count exchange, so its peers return an error instead of waiting.  Read the result with
 * trg_engine_export_csr(TRG_KIND_STITCHED).  (Replaces nothing in the reference: TRG has no multi-device
 * build) */
#define TRG_COMM_ID_BYTES 128
TrgStatus trg_engine_comm_unique_id(TrgEngine *e, uint8_t id[TRG_COMM_ID_BYTES]);
TrgStatus trg_engine_comm_init(TrgEngine *e, const uint8_t id[TRG_COMM_ID_BYTES], int32_t nranks, int32_t rank);
TrgStatus trg_engine_comm_adopt(TrgEngine *e, void *nccl_comm);
TrgStatus trg_engine_comm_destroy(TrgEngine *e);
TrgStatus trg_engine_stitch_exchange(TrgEngine *e, const float core_xyxy[4], int32_t cols, int32_t rows,
                                     int32_t *n_boundary, int32_t *n_cross);
/* why the last build fell back from the device path to the host replay ("" if it did not) */
const char *trg_engine_fallback_reason(const TrgEngine *e);

/* ---- instrumentation ------------------------------------------------------------------------- */
TrgStatus trg_engine_get_stats(const TrgEngine *e, TrgStats *out);
/* the direction table the engine uses (2^table_bits entries each) */
TrgStatus trg_engine_get_sampler_table(TrgEngine *e, float *cos_out, float *sin_out);
/* cell-sorted map arrays (n each), for index-build tests; any pointer may be NULL */
TrgStatus trg_engine_debug_map_index(TrgEngine *e, TrgKind map, float *x, float *y, float *z,
                                     int32_t *perm, int32_t *grid_wh, float *origin_cell);

#ifdef __cplusplus
}
#endif
#endif /* TRG_ENGINE_H_ */
