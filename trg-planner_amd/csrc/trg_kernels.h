// trg_kernels.h -- host-visible launch interface of the gfx950 kernels (trg_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace trg {

// Cell-sorted structure-of-arrays map index (replaces the reference's 2-D kd-tree of map points,
// trg.cpp:185-188 / kdtree.c).  Cells are row-major (cy * W + cx); the points of the cells
// [cx0..cx1] of one row are one contiguous, coalesced range of x[], y[], z[].
struct MapView {
  const float *x;
  const float *y;
  const float *z;
  const int *perm;        // original index of each sorted point (nearest-neighbour tie-break)
  const float4 *pt;       // the same points as 16-byte records (x, y, z, perm): what the tile staging reads --
                          // one load per point, and a row segment's ragged ends cost one partial line, not four
  const int *cell_start;  // W*H + 1 exclusive prefix of points per cell
  float x0, y0;           // grid origin (min x, min y of the cloud)
  float inv_g;            // 1 / cell size
  int W, H;
  int n;
};

// An accepted sample whose nearest map point was not unique (kd_nearest's visiting order decides
// which one the reference takes, trg.cpp:226-233 -> kdtree.c:303-362); the host resolves these.
constexpr int MAPTIE_CAP = 256;      // records per sampling launch
constexpr int MAPTIE_SET_CAP = 16;   // points listed per tie
struct MapTieRec {
  int slot;        // sample slot of the launch (node * S + j)
  float qx, qy;    // the sample position
  float d2;        // the tied fp32 squared distance
};
struct MapTieSet {
  int count;
  float d2;
  int sidx[MAPTIE_SET_CAP];   // index in the cell-sorted arrays
  int perm[MAPTIE_SET_CAP];   // original (insertion) index
  float x[MAPTIE_SET_CAP], y[MAPTIE_SET_CAP], z[MAPTIE_SET_CAP];
};

struct QueryParams {
  float robot_size;
  float height_threshold;
  float collision_threshold;
  float expand_dist;
  int sample_num;
  // tiled builds: samples outside [core_x0, core_x1) x [core_y0, core_y1) are rejected draws
  // (default: the whole plane, i.e. the reference behaviour)
  float core_x0, core_y0, core_x1, core_y1;
  // relative half-width of the band in which the device does not call the slope gate itself
  // (1e-4 by default; tests widen it to exercise the host decision path)
  float gate_margin;
};

// result codes of the position-only part of wireEdge (trg.cpp:269-363)
enum : int {
  EDGE_OK = 0,
  EDGE_GATE = 1,
  EDGE_SEG = 2,
  EDGE_EMPTY = 3,
  EDGE_FEW = 4,
  EDGE_STATUS_MASK = 7,
  EDGE_GATE_UNCERTAIN = 8,  // rational slope test too close to call: host decides with libm atan2f
  EDGE_CLAMPED = 16,        // weight < 0.1 -> 0 (trg.cpp:361-363)
};

// Instrumentation counters, sharded so that concurrent blocks do not serialise on one address:
// shard = blockIdx.x % COUNTER_SHARDS, one 128-byte line per shard, summed by the host.
constexpr int COUNTER_SHARDS = 64;
struct alignas(128) DeviceCounters {
  unsigned long long sample_hits;  // map points inside sample-collision discs
  unsigned long long edge_hits;    // map points inside segment discs + ellipse gathers (k_edges)
  unsigned long long spec_hits;    // same, speculative parent edges (k_spec_edges)
  unsigned long long nn_ties;
  unsigned long long overflow;     // disc queries that used the large-disc fallback
  unsigned long long pad[11];
};

// ---- index build -------------------------------------------------------------------------------
// bounds[4] = {min_x, min_y, max_x, max_y} as order-preserving uint32 keys; init with init_bounds
void launch_init_bounds(unsigned *d_bounds, hipStream_t s);
void launch_bounds(const float *d_xyz, size_t n, size_t stride, unsigned *d_bounds, hipStream_t s);
void launch_cell_count(const float *d_xyz, size_t n, size_t stride, float x0, float y0, float inv_g,
                       int W, int H, int *d_cell_of, int *d_rank, int *d_counts, hipStream_t s);
// exclusive scan of counts[0..m) into out[0..m], out[m] = total; tmp needs ceil(m/1024)+1 ints
// voxel-grid downsampling (trg_voxel.hip); *status 1 = leaf too small, input copied through
hipError_t voxel_grid_filter(const float *d_xyz, size_t n, size_t stride, float leaf, float *d_out,
                             size_t *n_out, int *status, hipStream_t s);
void launch_exclusive_scan(const int *d_counts, int *d_out, int m, int *d_tmp, hipStream_t s);
// index build through bins (trg_kernels.hip): plan = false when the grid has too many cells for it (the direct
// count / scatter path then); d_hist and d_base hold nbins * nwg + 1 ints, d_tmp (nbins * nwg) / 2048 + 4,
// the two scratch arrays n records each (scratch_a may be the map's own record array)
bool index_bins_plan(size_t n, size_t ncell, int *bin_shift, int *nbins, int *nwg);
void launch_index_bins(const float *d_xyz, size_t n, size_t stride, float x0, float y0, float inv_g, int W, int H,
                       int ncell, int bin_shift, int nbins, int nwg, int *d_hist, int *d_base, int *d_tmp,
                       float4 *d_scratch_a, float4 *d_scratch_b, int *d_cell_start, float *x, float *y, float *z,
                       int *perm, float4 *pt, hipStream_t s);
// scatter + per-cell sort through a scratch array of n 16-byte records (one store per point)
// (pt: the sorted points once more as 16-byte records, MapView::pt)
void launch_scatter_sort_aos(const float *d_xyz, size_t n, size_t stride, const int *d_cell_of,
                             const int *d_rank, int ncell, const int *d_cell_start, void *d_aos, float *x,
                             float *y, float *z, int *perm, float4 *pt, hipStream_t s);

// ---- queries -----------------------------------------------------------------------------------
void launch_probe_collision(const MapView &m, QueryParams p, float threshold, const float *d_xy,
                            int count, int *flag, int *cnt, int *n, DeviceCounters *ctr,
                            hipStream_t s);
void launch_probe_nearest_z(const MapView &m, QueryParams p, const float *d_xy, int count, float *z,
                            int *found, DeviceCounters *ctr, hipStream_t s);
// p1/p2: count x 3 floats
// mid: device scratch of edge_mid_floats(count) floats (phase-1 records consumed by phase 2)
void launch_edges(const MapView &m, QueryParams p, const float *d_p1, const float *d_p2, int count,
                  float *mid, int *status, int *n_pts, float *weight, float *dist,
                  DeviceCounters *ctr, hipStream_t s);
size_t edge_mid_floats(size_t edges);

// Expansion of `count` queued nodes (trg.cpp:384-403 sampling + the elevation lookup :244-247):
//   node_xy[count*2], node_id[count] (sampler key), outputs per node: n_acc, n_draws and
//   per slot (node*S + j): sample x, y, z
void launch_sample_nodes(const MapView &m, QueryParams p, const float *cos_t, const float *sin_t,
                         int table_bits, uint32_t seed, uint32_t epoch, const float *node_xy,
                         const int *node_id, int count, int *n_acc, int *n_draws, float *sx,
                         float *sy, float *sz, DeviceCounters *ctr, int *mt_count,
                         MapTieRec *mt_rec, hipStream_t s);
// exact nearest-point tie-break helpers (rare path, see map_nn_exact in trg_engine.cpp)
void launch_map_tied_set(const MapView &m, float qx, float qy, float r0, MapTieSet *d_out,
                         hipStream_t s);
// State of one walk down the map's insertion tree (which of two tied points kd_nearest visits first):
// the steps run on the device back to back, the host only looks at the final state.
struct MapTieWalk {
  unsigned long long key;  // smallest (original index << 32 | sorted index) of the current scan; ~0 = none
  float lo[2], hi[2];      // half-open region of the current subtree
  int cur_perm;            // its ancestor (points inserted later than this one are in the subtree)
  int axis;
  float qx, qy;
  int aperm, bperm;
  float ax, ay, bx, by;
  int done;                // 0 walking, 1 decided (first = 0: A, 1: B), 2 lost its candidates
  int first;
  int steps;
  int ticket;              // workgroups of the current grid step that have finished
};
void launch_map_tie_walk(const MapView &m, MapTieWalk *d_state, int grid_steps, int block_steps, hipStream_t s);
// out_xy[2 * k], [2 * k + 1] = position of the point with original index k, for k < M
void launch_collect_first(const MapView &m, int M, float *d_out_xy, hipStream_t s);
// Speculative parent edges node -> sample for every accepted sample of a chunk:
//   node_xyz[count*3]; slot = node*S + j evaluated iff j < n_acc[node]
void launch_spec_edges(const MapView &m, QueryParams p, const float *node_xyz, int count,
                       const int *n_acc, const float *sx, const float *sy, const float *sz,
                       float *mid, int *status, int *n_pts, float *weight, float *dist,
                       DeviceCounters *ctr, hipStream_t s);

// ---- device-resident BFS (trg_bfs.inc, trg_level.inc) -------------------------------------------------
constexpr int GRID_SLOTS = 4;      // nodes per grid cell (cell = robot_size, nodes are >= robot_size apart)
constexpr int BFS_TIE_CAP = 64;    // tied slots of one level handed to the host one by one (more: the level is replayed)
constexpr int BFS_UNC_CAP = 4096;  // uncertain slope gates handed to the host per sync point
enum : int {
  BFS_CTR_V = 0, BFS_CTR_MNEXT = 1, BFS_CTR_NCAND = 2, BFS_CTR_NUNC = 3, BFS_CTR_ERR = 4,
  BFS_CTR_DONE = 5,   // in the host copy: the stamp of the level (as word 15)
  BFS_CTR_NTIE = 5,   // on the device: entries of tie_list (slots whose nearest node was not unique)
  BFS_CTR_NMAPTIE = 6 /* and 7: one list per level parity */,
  BFS_CTR_NUNC1 = 8,  // uncertain gates of odd levels (the next level is expanded while the host
                      // still looks at this one)
  // (9..13 spare)
  BFS_CTR_NUNC2 = 14,      // uncertain gates of the deferred evaluations (third list)
  BFS_CTR_COUNT = 16
};
enum : int {
  BFS_ERR_GRID_OVERFLOW = 1, BFS_ERR_NB_OVERFLOW = 2, BFS_ERR_VCAP = 4, BFS_ERR_TIE = 8,
  BFS_ERR_HASH = 16, BFS_ERR_LEVEL_TOO_BIG = 32, BFS_ERR_CLEAN = 64,
  BFS_ERR_STALL = 128,  // k_level_resolve's bounded wait ran out: the level is replayed on the host
  BFS_ERR_LOOKBACK = 512,  // the commit's look-back scan ran out of polls (another process on the card): the
                           // level's numbering is void, the whole build is redone by the host replay
  BFS_ERR_STEP3 = 1024,  // step 3: more neighbours than a list / the call log's stride holds, or a slope gate of a
                         // rescue edge the device could not call: the build goes to the host replay
  BFS_ERR_TIE_CLS = 256 // a sample's nearest PRE-LEVEL node was not unique (k_level_sample; the slot
                        // carries SLOT_TIE): the host picks the reference's winner, resolve + commit rerun
};
enum : int { CALL_NONE = -2, CALL_PENDING = -1 };

// One cell of the node hash grid (cell = robot_size): everything a probe needs in one 64-byte line.
struct alignas(64) GridCell {
  int cnt;
  int id[GRID_SLOTS];
  float x[GRID_SLOTS], y[GRID_SLOTS];
  int pad[3];
};
// Everything the level kernels keep per sample slot (slot = queue position * S + j), 64 bytes.
struct alignas(16) SlotRec {
  float x, y, z;   // accepted sample: position and elevation of the nearest map point
  float d0sq;      // squared distance to the nearest node that existed before the level
  int nn0;         // that node (-1: none in reach)
  int cls;         // low byte: 0 unused slot, 1 a pre-level node within robot_size, 2 candidate for a
                   // new node; SLOT_TIE: the nearest pre-level node was not unique
  int status;      // speculative parent edge (candidates): EDGE_* code and flags
  float dist;      //   its length
  float cov[6];    //   covariance of its gather (xx xy xz yy yz zz), or cov[0] = weight when w_given
  int w_given;     //   1: the weight itself is stored (host re-evaluation after a map tie)
  int hits;        //   map points inside the edge's query radii (instrumentation)
};
enum : int { SLOT_CLS_MASK = 0xFF, SLOT_TIE = 0x100 };
struct alignas(16) NodeRec {  // per queued node of a level, 48 bytes
  int n_acc, n_draws;         // accepted samples, draws made
  int hits_sample, hits_spec; // map points inside its sampling discs / speculative-edge queries
  unsigned cand_lo, cand_hi;  // bit j: accepted sample j is a candidate
  int stamp, id;              // set by a p_role workgroup of the previous level's resolve launch: the level tag
                              // and node id its samples (n_acc, n_draws, hits_sample, slot x y z) belong to
  float px, py, pz;           // the node itself (k_level_spec: its speculative edges start here)
  int pad2;
};
struct alignas(16) HashEnt {  // candidate hash of a level: node-grid cell -> candidate
  unsigned long long tagcell; // level tag << 32 | cell; entries of other levels count as empty
  int slot;
  int pad;
  float x, y;
  int pad2[2];
};
// expandGraph's step 3 inside a level (trg_step3.inc): who can make a candidate whose parent edge failed a
// valid node after all.  Per sample slot; written for such candidates only.
constexpr int RESC_MAX = 14;
struct alignas(16) RescueRec {
  int pre;             // 1: an edge to a node that existed before the level succeeds
  int n;               // earlier candidates of the level whose edge from this sample succeeds ...
  int slot[RESC_MAX];  // ... their slots
};
struct alignas(16) NodeCov {  // what the edge to a node created by the BFS still needs: its weight
  float cov[6];               // is computed after the level loop (k_node_weights)
  int w_given;
  int call;                   // the wireEdge call that created the node
};

struct BfsDev {
  // node store (creation order) and its hash grid
  float *nx, *ny, *nz;
  int *nstate;
  NodeCov *ncov;
  int vcap;
  GridCell *gcell;
  float gx0, gy0, ginv, gcell_size;
  int GW, GH;
  // frontier ping-pong
  int *front_cur, *front_next;
  float2 *fxy_cur, *fxy_next;  // their positions (the sampling workgroup starts from here)
  int fcap;
  // level records (two sets, by level parity)
  NodeRec *node_rec;
  SlotRec *slot_rec;
  int *c_outcome;   // per slot: resolve outcome (polled across workgroups)
  unsigned long long *c_cell;  // per slot of the level being resolved, resolve -> commit workgroups: launch epoch << 44 |
                               // creates a node (0 no, 1 Invalid, 2 valid) << 42 | node-grid cell * GRID_SLOTS + its place
  HashEnt *lv_hash;
  int ht_size;
  RescueRec *resc;   // step-3 builds only (else null)
  unsigned long long *wg_state;  // per resolve workgroup: epoch | state | created | valid (look-back scan of the commit)
  unsigned *ticket;  // start-order tickets of the resolve workgroups (runs on across launches)
  unsigned long long *front_ready;  // [fcap + 1] handshake words commit -> p_role workgroups: entry b of the next
                                    // frontier is numbered (tag | creating slot | id); [fcap]: its length
  int4 *nexp;       // per expanded node: accepted | candidates << 8, draws, disc hits, speculative-edge hits
  int *nhits;       // per created node: map points inside its parent edge's queries
  // deferred edge evaluation scratch
  float *mid;
  // uncertain slope gates for the host (2 x BFS_UNC_CAP, by level parity)
  int *unc_list;
  float *unc_rec;
  int *tie_list;      // BFS_TIE_CAP slots of the current level that met an exact distance tie in k_level_resolve
  MapTieRec *mt_rec;  // 2 x MAPTIE_CAP records (level parity), counts in ctrs[BFS_CTR_NMAPTIE + parity]
  // call log (one record per sample slot, in program order)
  int *call_n1, *call_n2, *call_status;
  float *call_w, *call_dist;
  int *newid_of_call;  // node id created by call number i (set by the commit in k_level_resolve)
  int cstride;         // call-log entries per sample slot: 1, or LEVEL_STEP3_STRIDE when expandGraph's step 3 is on
  // counters
  int *ctrs;
  int *host_ctrs;      // pinned host copy of ctrs + stamp, written by the next level's k_level_sample (may be null)
  unsigned long long *stats64;  // [6]: longest resolve wait; [8..13]: expand phase cycles (profiling)
  unsigned long long *tl;       // wall-clock marks of a few levels (-DLV_TIMELINE builds with TRG_TIMELINE set; else null)
};

struct FinDev {
  unsigned long long *ht_key;
  int *ht_seq;
  int *ok_seq;  // per table entry: the smallest call number among the pair's SUCCESSFUL calls (its edge)
  unsigned ht_size;
  int *call_slot;
  int *deg, *fill, *rowptr;
  int *col, *seq;
  float *w, *dist;
};

void launch_bfs_insert_nodes(const BfsDev &B, int first, int count, hipStream_t s);
void launch_bfs_clear_tie(const BfsDev &B, hipStream_t s);
// takes the grid reservations of a level's created nodes back (from the slots' outcomes)
void launch_bfs_undo_slots(const BfsDev &B, int slots, hipStream_t s);
// ---- one BFS level in three kernels (trg_level.inc) ---------------------------------------------
constexpr int LEVEL_MAX_SAMPLES = 64;  // sample_num the level kernels support
constexpr int LEVEL_STEP3_STRIDE = 16; // call-log entries per slot with step 3 on: the slot's own call + up to 15 neighbour calls
// expansion of the frontier nodes [node_base, count) (count_dev != nullptr: count is an upper
// bound, the kernel takes min(count, *count_dev)); tag: hash tag of this level attempt (>= 1);
// pub_stamp != 0: the launch first hands B.ctrs (the counters of the level committed just before
// it in the stream) to the host: B.host_ctrs[0..15], words 5 and 15 = pub_stamp
void launch_level_expand(const MapView &m, QueryParams p, const float *cos_t, const float *sin_t,
                         int table_bits, uint32_t seed, uint32_t epoch, const BfsDev &B, int count,
                         const int *count_dev, int node_base, int parity, int tag, int pub_stamp,
                         DeviceCounters *ctr, hipStream_t s, int which = 3);
// whether the level kernels can serve these parameters (window of the node grid, sample count)
bool level_kernels_support(const QueryParams &p, float grid_cell);
// The pure part of the NEXT level's expansion, run by trailing workgroups of the resolve launch for the
// frontier entries [0, count) as the commit numbers them (count = 0: none)
struct LevelNext {
  void *node_rec = nullptr, *slot_rec = nullptr;  // the next level's record set
  int count = 0, parity = 0, tag = 0;
  const float *cos_t = nullptr, *sin_t = nullptr;
  int table_bits = 0;
  uint32_t seed = 0, epoch = 0;
  DeviceCounters *ctr = nullptr;
};
// ticketed: the workgroups take their logical index from a start ticket (*B.ticket; ticket_base = tickets
// drawn by earlier launches, advanced here) instead of blockIdx.x -- the repeat of a launch whose wait ran out
void launch_level_resolve_commit(const MapView &m, const BfsDev &B, QueryParams p, int count, int new_state,
                                 long long call_base, int V0, int tag, int epoch, hipStream_t s,
                                 int stall_test,  // 0; test hooks: 1 a candidate stays undecided, 2 a look-back gives up
                                 bool ticketed, unsigned *ticket_base, const LevelNext &next);
// sums of the per-node expansion statistics into out[0..5] (added to what is there)
typedef int TrgStatus_t;  // 0 ok
// step 3 after the level loop: the reference's node tree rebuilt on the device (creation order), then the
// neighbour calls of every valid node into the call log behind its creating call; scratch: 6 V + 8 ints
TrgStatus_t launch_step3_calls(const BfsDev &B, int V, float range, int *scratch, int *h_left, hipStream_t s);
void launch_bfs_stats(const BfsDev &B, int V, unsigned long long *out, hipStream_t s);
// weights of the edges to the nodes [1, V) the BFS created (covariance -> SVD -> weight)
void launch_node_weights(const BfsDev &B, int V, hipStream_t s);
// deferred wireEdge evaluations of the calls list[0..count) (count_dev != nullptr: count is an upper
// bound, the kernels take min(count, *count_dev))
void launch_calls_eval(const MapView &m, QueryParams p, const BfsDev &B, const int *list, int count,
                       const int *count_dev, DeviceCounters *ctr, hipStream_t s);
void launch_first_insert(const FinDev &F, const BfsDev &B, long long c0, long long c1, hipStream_t s);
void launch_calls_select(const FinDev &F, const BfsDev &B, long long c0, long long c1, int round,
                         int *flag, int *off, int *scan_tmp, int *list, unsigned long long *total,
                         hipStream_t s, int *count_out = nullptr);
// round 2 over the whole log in one pass (*count must be zero): calls whose pair's first call failed
void launch_calls_select2_append(const FinDev &F, const BfsDev &B, long long ncalls, int *list, int *count,
                                 unsigned long long *total, hipStream_t s);
void launch_fin_insert_count(const FinDev &F, const BfsDev &B, long long ncalls, hipStream_t s);
void launch_fin_scatter(const FinDev &F, const BfsDev &B, long long ncalls, hipStream_t s);
void launch_fin_rowsort(const FinDev &F, int V, hipStream_t s);
void launch_fin_clean(const FinDev &F, const BfsDev &B, const int *map_order, int V, int *keep_flag,
                      int *keep_pos, int *new2old, int *old2new, int *deg_new, int *rowptr_new,
                      int *scan_tmp, int *col, float *w, float *dist, float *xyz, int *state,
                      hipStream_t s);

// ---- tile-boundary stitch of the tiled build (trg_stitch.inc) ----------------------------------------
struct StitchRec {  // a boundary node: local id and position (16 bytes; layout of TrgBoundaryRec)
  int id;
  float x, y, z;
};
struct StitchEdge {  // a cross edge (24 bytes; layout of TrgCrossEdge)
  int tile_a, id_a, tile_b, id_b;
  float weight, dist;
};
constexpr int STITCH_MAX_TILES = 64;
// sides: bit 0 left, 1 right, 2 bottom, 3 top side of the core has a neighbouring tile
void launch_stitch_boundary(const float *d_xyz, int V, const float core[4], int sides, float d, int *flag,
                            int *off, int *scan_tmp, StitchRec *rec, int cap, hipStream_t s);
void launch_stitch_pairs(const StitchRec *all, const int *rec_offsets, int tile, int ntiles, float d,
                         int pass, int *cnt, const int *off, int *pair_a, int *pair_b, hipStream_t s);
void launch_stitch_pair_points(const StitchRec *all, const int *pair_a, const int *pair_b, int n,
                               float *p1, float *p2, hipStream_t s);
void launch_stitch_edge_select(const StitchRec *all, const int *rec_offsets, int tile, int ntiles,
                               const int *pair_a, const int *pair_b, const int *status,
                               const float *weight, const float *dist, int n, int *flag, int *off,
                               int *scan_tmp, int *n_unc, StitchEdge *out, int cap, hipStream_t s);
void launch_stitch_assemble(const StitchEdge *edges, int n_edges, int tile, int ntiles,
                            const int *node_offsets, const int *rowptr, const int *col, const float *w,
                            const float *dist, int V, int *extra, int *deg, int *rowptr_new,
                            int *scan_tmp, int *fill, int *col_new, float *w_new, float *dist_new,
                            int phase, hipStream_t s);

}  // namespace trg
