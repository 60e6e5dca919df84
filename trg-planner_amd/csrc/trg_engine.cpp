// trg_engine.cpp -- host side of the MI355X TRG construction engine and its C ABI
// (include/trg_engine.h).
//
// Division of labour:
//   GPU  : every function of (positions, map) -- map index build, isCollision, elevation lookup,
//          rejection sampling, edge risk (segment walk + ellipse gather + PCA).  trg_kernels.hip
//   host : everything that depends on graph STATE and is inherently sequential in the reference --
//          the FIFO of expandGraph (trg.cpp:376-381), nearest existing node / merge test
//          (trg.cpp:408-417), wireEdge's dedupe (trg.cpp:255-267), cleanGraph's renumbering
//          (trg.cpp:491-535), A* (trg.cpp:603-690).  The host never evaluates a map query itself:
//          without a working HIP device every entry point fails with TRG_ERR_DEVICE.
//
// The BFS is replayed in exactly the reference's order; GPU work is issued ahead of the replay in
// chunks of queued nodes (the samples of a node depend only on its position and id), so device
// latency hides behind the host loop.
#include <hip/hip_runtime.h>
#include <math.h>  // float overloads of atan2 etc. in the global namespace, as the reference has

#include <algorithm>
#include <array>
#include <atomic>
#include <cstddef>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <queue>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/trg_engine.h"
#include "graph_json.h"
#include "host_index.h"
#include "map_order_sim.h"
#include "trg_kernels.h"

namespace {

using namespace trg;
using Clock = std::chrono::steady_clock;

inline double ms_since(Clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
}
inline float norm2f(float dx, float dy) { return sqrtf(dx * dx + dy * dy); }

// fn(begin, end) over [0, n) on up to 8 host threads (bulk host passes over the graph: cleanGraph, CSR <-> edge pool)
template <typename F>
void parallel_ranges(size_t n, F fn) {
  const size_t hw = std::max(1u, std::thread::hardware_concurrency());
  const size_t T = std::min<size_t>(std::min<size_t>(8, hw), (n + 65535) / 65536);
  if (T <= 1) {
    fn((size_t)0, n);
    return;
  }
  std::vector<std::thread> thr;
  for (size_t t = 1; t < T; ++t) thr.emplace_back(fn, n * t / T, n * (t + 1) / T);
  fn((size_t)0, n / T);
  for (auto &th : thr) th.join();
}

struct IndexScratch {  // temporaries of build_index, grown on demand
  int *cell_of = nullptr, *rank = nullptr, *counts = nullptr, *tmp = nullptr;  // the direct path (huge grids only)
  int *hist = nullptr, *base = nullptr, *bin_tmp = nullptr;                    // the path through bins
  void *aos = nullptr;
  size_t cap_pts = 0, cap_cells = 0, cap_direct = 0, cap_bins = 0;
};

struct DevMap {
  size_t n = 0;
  float *x = nullptr, *y = nullptr, *z = nullptr;
  int *perm = nullptr, *cell_start = nullptr;
  float4 *pt = nullptr;  // the sorted points as 16-byte records (MapView::pt)
  size_t cap_pts = 0, cap_cells = 0;
  MapView view{};
  float g = 0;
  float bounds[4] = {0, 0, 0, 0};
  bool valid = false;
  // the top of the reference's insertion-built kd-tree over this map: its first points (original index
  // < top_m), built lazily when a nearest-point tie has to be broken (map_first_of_two)
  int top_m = 0;
  std::vector<float> top_xy;
  std::vector<int> top_left, top_right;
  // For the global map the top is prepared beside the build (start_map_top): its points are fetched on the aux
  // stream and a helper thread inserts them; whoever needs the top (or replaces / frees the map) joins it first.
  std::thread top_thread;
  void top_wait() {
    if (top_thread.joinable()) top_thread.join();
  }
};

template <typename T>
struct PinnedBuf {
  T *h = nullptr;
  T *d = nullptr;
  size_t cap = 0;
};

// results of one speculative / deferred edge evaluation as the replay consumes them
struct CallRec {
  int n1, n2;
  int status;  // EDGE_* | flags, -1 = pending
  float weight, dist;
};

// std::vector whose resize() leaves trivially constructible elements uninitialised (the bulk builders overwrite
// every element; value-initialising four 6.7 M-entry arrays cost 10 ms per cleanGraph at C3)
template <typename T>
struct NoInitAlloc : std::allocator<T> {
  template <typename U>
  struct rebind {
    using other = NoInitAlloc<U>;
  };
  NoInitAlloc() = default;
  template <typename U>
  NoInitAlloc(const NoInitAlloc<U> &) {}
  template <typename U>
  void construct(U *p) noexcept {
    ::new ((void *)p) U;
  }
  template <typename U, typename... A>
  void construct(U *p, A &&...a) {
    ::new ((void *)p) U(std::forward<A>(a)...);
  }
};
template <typename T>
using RawVec = std::vector<T, NoInitAlloc<T>>;

struct EdgePool {
  RawVec<int> dst, next;
  RawVec<float> w, dist;
  std::vector<int> head, tail, deg;
  void reset(size_t nodes) {
    dst.clear();
    next.clear();
    w.clear();
    dist.clear();
    head.assign(nodes, -1);
    tail.assign(nodes, -1);
    deg.assign(nodes, 0);
  }
  void grow_nodes(size_t nodes) {
    if (head.size() < nodes) {
      head.resize(nodes, -1);
      tail.resize(nodes, -1);
      deg.resize(nodes, 0);
    }
  }
  bool has(int a, int b) const {
    for (int e = head[a]; e >= 0; e = next[e])
      if (dst[e] == b) return true;
    return false;
  }
  // rows laid out one after the other (offs[nodes + 1]); the caller fills dst / w / dist of every row and calls
  // link_rows: the lists then read like pushes in that order, and later pushes append behind them
  void alloc_rows(const std::vector<int> &offs) {
    const size_t nodes = offs.size() - 1, E = (size_t)offs[nodes];
    const size_t room = E + E / 8 + 65536;  // (the pushes that follow must not reallocate 100 MB)
    dst.reserve(room);
    next.reserve(room);
    w.reserve(room);
    dist.reserve(room);
    dst.resize(E);
    next.resize(E);
    w.resize(E);
    dist.resize(E);
    head.resize(nodes);
    tail.resize(nodes);
    deg.resize(nodes);
  }
  void link_rows(const std::vector<int> &offs, size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; ++i) {
      const int a = offs[i], b = offs[i + 1];
      deg[i] = b - a;
      head[i] = b > a ? a : -1;
      tail[i] = b > a ? b - 1 : -1;
      for (int k = a; k < b; ++k) next[k] = k + 1 < b ? k + 1 : -1;
    }
  }
  void push(int a, int b, float ww, float dd) {
    const int e = (int)dst.size();
    dst.push_back(b);
    w.push_back(ww);
    dist.push_back(dd);
    next.push_back(-1);
    if (tail[a] >= 0) {
      next[tail[a]] = e;
    } else {
      head[a] = e;
    }
    tail[a] = e;
    deg[a]++;
  }
};

// Minimal vector over pinned host memory (hipHostMalloc): the CSR the engine hands out is the
// destination of the final device-to-host copies, which run at full PCIe rate only into pinned
// pages; no value-initialisation on resize (an 80 MB memset would cost as much as the copy).
template <typename T>
struct PVec {
  T *p = nullptr;
  size_t n = 0, cap = 0;
  bool pinned = false;
  PVec() = default;
  PVec(const PVec &) = delete;
  PVec &operator=(const PVec &o) {
    resize(o.n);
    if (o.n) memcpy(p, o.p, o.n * sizeof(T));
    return *this;
  }
  PVec &operator=(const std::vector<T> &o) {
    resize(o.size());
    if (!o.empty()) memcpy(p, o.data(), o.size() * sizeof(T));
    return *this;
  }
  ~PVec() { release(); }
  void release() {
    if (p) {
      if (pinned) (void)hipHostFree(p);
      else free(p);
    }
    p = nullptr;
    n = cap = 0;
  }
  void reserve(size_t m) {
    if (m <= cap) return;
    size_t nc = std::max(m, cap + cap / 2);
    T *np = nullptr;
    bool pin = hipHostMalloc((void **)&np, nc * sizeof(T), hipHostMallocDefault) == hipSuccess;
    if (!pin) np = (T *)malloc(nc * sizeof(T));
    if (n) memcpy(np, p, n * sizeof(T));
    if (p) {
      if (pinned) (void)hipHostFree(p);
      else free(p);
    }
    p = np;
    cap = nc;
    pinned = pin;
  }
  void resize(size_t m) {
    reserve(m);
    n = m;
  }
  void assign(size_t m, const T &v) {
    resize(m);
    for (size_t i = 0; i < m; ++i) p[i] = v;
  }
  void clear() { n = 0; }
  bool empty() const { return n == 0; }
  size_t size() const { return n; }
  T *data() { return p; }
  const T *data() const { return p; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
  T *begin() { return p; }
  T *end() { return p + n; }
  const T *begin() const { return p; }
  const T *end() const { return p + n; }
  void push_back(const T &v) {
    if (n == cap) reserve(std::max<size_t>(16, cap * 2));
    p[n++] = v;
  }
  template <typename It>
  void append(It first, It last) {
    const size_t m = (size_t)(last - first);
    reserve(n + m);
    for (size_t i = 0; i < m; ++i) p[n + i] = first[i];
    n += m;
  }
};

struct Csr {
  PVec<float> xyz;
  PVec<int32_t> state, rowptr, col, cid;
  PVec<float> w, dist;
  void clear() {
    xyz.clear();
    state.clear();
    rowptr.clear();
    col.clear();
    cid.clear();
    w.clear();
    dist.clear();
  }
};

template <typename T>
struct View {  // host / device views into a chunk's packed blobs
  T *h = nullptr;
  T *d = nullptr;
};

struct Chunk {
  int first = 0, count = 0;  // queue positions [first, first+count)
  bool in_flight = false;
  hipEvent_t done = nullptr;
  hipEvent_t t0 = nullptr, t1 = nullptr, t2 = nullptr;  // sample start / sample end / edges end
  // one packed input blob (H2D) and one packed output blob (D2H) per chunk: a single copy each way
  PinnedBuf<uint32_t> in_blob, out_blob;
  float *d_mid = nullptr;  // phase-1 edge records (device only)
  size_t mid_cap = 0;
  View<float> node_xy, node_xyz, sx, sy, sz, weight, dist;
  View<int> node_id, n_acc, n_draws, status;
  // nearest-map-point ties of accepted samples: [0] = count, records from word 4 (MapTieRec)
  PinnedBuf<int> mt;
  size_t in_words = 0, out_words = 0;
  void carve(int cnt, int S) {
    const size_t c = (size_t)cnt, cs = c * (size_t)S;
    auto iv = [&](size_t off) { return View<int>{(int *)in_blob.h + off, (int *)in_blob.d + off}; };
    auto fv = [&](size_t off) { return View<float>{(float *)in_blob.h + off, (float *)in_blob.d + off}; };
    node_xy = fv(0);
    node_xyz = fv(2 * c);
    node_id = iv(5 * c);
    in_words = 6 * c;
    auto io = [&](size_t off) { return View<int>{(int *)out_blob.h + off, (int *)out_blob.d + off}; };
    auto fo = [&](size_t off) { return View<float>{(float *)out_blob.h + off, (float *)out_blob.d + off}; };
    n_acc = io(0);
    n_draws = io(c);
    sx = fo(2 * c);
    sy = fo(2 * c + cs);
    sz = fo(2 * c + 2 * cs);
    status = io(2 * c + 3 * cs);
    weight = fo(2 * c + 4 * cs);
    dist = fo(2 * c + 5 * cs);
    out_words = 2 * c + 6 * cs;
  }
};

struct EdgeBatch {
  bool in_flight = false;
  int count = 0;
  hipEvent_t done = nullptr, t0 = nullptr, t1 = nullptr;
  std::vector<int> call_idx;  // which CallRec each row fills
  PinnedBuf<float> p1, p2, weight, dist;
  PinnedBuf<int> status;
  float *d_mid = nullptr;
};

struct BfsBuffers;  // device-resident BFS state (trg_engine_bfs.inc)
struct StitchBufs;  // scratch of the tile-boundary stitch (trg_engine_stitch.inc)
struct Uploader;    // host cloud -> HBM staging (upload_and_build)
struct ExchangeState;  // RCCL communicator + buffers of the native stitch exchange (trg_engine_exchange.inc)

}  // namespace

struct TrgEngine;
namespace {
TrgStatus stitch_fetch(TrgEngine *e);  // trg_engine_stitch.inc
void exchange_release(TrgEngine *e);   // trg_engine_exchange.inc
}
// Scratch of one A* search over the CSR.  open_check / close_list of the reference (trg.cpp:619-620,
// unordered_maps keyed by node id) are flat arrays whose entries count only when their stamp equals the
// search's generation, so nothing of size V is cleared per query.
struct PlanScratch {
  struct Opt {  // OptimizeNode (TRG.h:42-48): f_cost and g_cost are floats
    int id;
    int parent;  // index into pool, -1 for the start
    float f, g;
  };
  std::vector<Opt> pool;
  std::vector<int> heap;
  std::vector<uint32_t> open_gen, close_gen;
  std::vector<int> open_idx;
  uint32_t gen = 0;
  std::vector<int> chain, hits, tied;
  void begin(size_t V) {
    if (open_gen.size() != V) {
      open_gen.assign(V, 0);
      close_gen.assign(V, 0);
      open_idx.assign(V, -1);
      gen = 0;
    }
    if (++gen == 0) {  // wrapped: start over
      std::fill(open_gen.begin(), open_gen.end(), 0u);
      std::fill(close_gen.begin(), close_gen.end(), 0u);
      gen = 1;
    }
    pool.clear();
    heap.clear();
  }
};

struct TrgEngine {
  TrgParams prm{};
  int device = 0;
  std::string err;
  std::string arch;
  bool device_ok = false;

  hipStream_t s_main = nullptr, s_edge = nullptr;
  hipStream_t s_aux = nullptr;  // rare-event work (nearest-point tie walks) beside whatever the main stream holds
  DevMap gmap, lmap;
  IndexScratch idx_scratch;
  float *top_xy_d = nullptr, *top_xy_h = nullptr;  // staging of start_map_top (device / pinned host)
  hipEvent_t top_ev = nullptr;
  DeviceCounters *d_ctr = nullptr;
  unsigned *d_bounds = nullptr;

  // sampler
  TrgSampler sampler{1, 16};
  std::vector<float> cos_t, sin_t;
  float *d_cos = nullptr, *d_sin = nullptr;
  int table_bits_dev = 0;
  uint32_t epoch = 0;
  uint32_t epoch_base = 0;  // tiled builds: sampler epoch of this tile
  float core[4] = {-INFINITY, -INFINITY, INFINITY, INFINITY};  // tiled builds: node creation region

  // graph state: slot == id (ids are dense at all times)
  std::vector<float> nx, ny, nz;
  std::vector<int> nstate;
  std::vector<int> ncid;  // creation index inside the last build
  std::unordered_map<int, int> order_map;  // mirrors trgStruct::nodes (iteration order only)
  // After a device build the container history is carried by an O(n) replica of the hashtable's
  // iteration order (map_order_sim.h); the real map is rebuilt from it only if a host path needs it.
  MapOrderSim nodes_sim;
  bool real_map_stale = false;
  std::vector<int> last_new2old;  // of the last host cleanGraph: old id of every surviving node
  EdgePool edges;
  int node_id = 0;
  float root_pos[2] = {0, 0};
  float local_root[2] = {0, 0};
  std::unordered_map<int, int> local_map;  // mirrors local trgStruct::nodes (iteration order)
  std::vector<int> local_nodes;  // ids in local_map's iteration order
  NodeGrid grid;
  NodeKd kd;          // reference-shaped index of the CURRENT global node set
  bool kd_valid = false;
  std::vector<int> kd_insert_order;  // ids in the order the reference inserted them
  NodeKd lkd;         // local node tree (isFrontier does not use it; kept for completeness)
  bool step3 = false;

  // build scratch
  std::vector<int> queue;
  std::vector<CallRec> calls;
  static constexpr int NCHUNK = 3;
  static constexpr int CHUNK_MAX = 4096;
  Chunk chunks[NCHUNK];
  Chunk root_chunk;  // samples + speculative edges of updateGraph's expansion roots, fetched in bulk
  static constexpr int NEBATCH = 3;
  static constexpr int EBATCH_MAX = 1 << 16;
  EdgeBatch ebatches[NEBATCH];
  std::vector<int> pending_calls;
  int chunk_S = 0;

  // small synchronous scratch
  PinnedBuf<float> sy_in, sy_in2, sy_f0, sy_f1;
  PinnedBuf<int> sy_i0, sy_i1, sy_i2;
  float *sy_mid = nullptr;
  size_t sy_cap = 0;
  // exact nearest-map-point tie-break scratch (map_nn_exact)
  MapTieSet *mt_set_d = nullptr, *mt_set_h = nullptr;
  MapTieWalk *mt_walk_d = nullptr, *mt_walk_h = nullptr;

  Csr csr_global, csr_pre, csr_local;
  Csr csr_stitched;              // tiled builds: this tile's rows of the stitched global graph
  bool stitched_on_device = false;  // ... their edge arrays are still in HBM only (fetched on export)
  int stitched_edges = 0;
  bool dev_csr_valid = false;    // the cleaned global CSR of the last device build is still in HBM
  StitchBufs *stitch = nullptr;
  bool keep_preclean = false;    // instrumentation: snapshot the graph before cleanGraph
  bool use_device_bfs = true;    // device-resident BFS when expandGraph's step 3 is disabled
  int defer_overlap = 1;         // 1: deferred edge evaluations pipelined behind the level loop on a 2nd stream;
                                 // 2: only the pair-table inserts + first-of-pair selection run beside the loop
                                 // (measured: the loop loses more than the pipeline gains; kept as an option)
  uint64_t graph_version = 1;    // bumped by everything that changes the global graph (queries cache per version)
  // planner state (A* runs straight on csr_global, see plan_on_csr): positions in the node tree's insertion
  // order for the rare order questions (nearest-node ties, several goal hits), and the search scratch
  std::vector<float> kdo_x, kdo_y;
  std::vector<int> kdo_index;    // node id -> position in the insertion order
  uint64_t kdo_version = 0;
  PlanScratch *plan_scratch = nullptr;
  Uploader *uploader = nullptr;
  ExchangeState *exchange = nullptr;  // host cloud -> HBM staging (upload_and_build)
  bool pool_valid = true;        // e->edges mirrors csr_global
  bool host_grid_valid = true;   // e->grid holds the current node set
  bool kd_order_dirty = false;   // kd_insert_order must be re-derived from order_map
  int debug_tie_every = 0;       // test hook: treat every n-th BFS level as tie-affected
  int debug_spec_bound = 0;      // test hook: cap the speculative sampling launch at n nodes
  int debug_fallback_level = -1; // test hook: the device BFS declines at this level
  bool tie_inplace = true;       // node-distance ties settled slot by slot on the committed level (off: host level replay)
  int debug_lookback_level = -1; // test hook: one workgroup's commit look-back gives up at this level
  int debug_stall_level = -1;    // test hook: k_level_resolve leaves one candidate of this level undecided
  bool debug_wait_rerun = false; // test hook: the ticketed repeat of that launch runs into the same hook
  bool debug_call_stride = false; // test hook: the sparse call log of step-3 builds on any configuration
  bool step3_device = true;      // configurations with expandGraph's step 3 on the device-resident path too (off: host replay)
  bool presample = false;        // pure part of the next level's expansion inside the resolve launch (p_role workgroups):
                                 // measured +3 ms per C3 build (the 8-wave workgroups hold the places the resolve workgroups free)
  int resolve_tickets = 0;       // 1: every resolve launch takes its workgroup indices from start tickets (default: only
                                 // the repeat of a launch whose bounded wait ran out)
  float gate_margin = 1e-4f;     // band in which the slope gate is left to the host's libm
  BfsBuffers *bfs = nullptr;
  std::string bfs_fallback_reason;
  // map points inside the queries of the level kernels of the last device build (instrumentation;
  // counted per committed level, so discarded launches count nothing)
  uint64_t lv_hits_sample = 0, lv_hits_spec = 0;
  TrgStats stats{};

  // goal state (trg.h:121-126)
  int goal_node = -1;
  bool goal_known = false;
  float goal_pose2d[2] = {0, 0};

  TrgStatus fail(TrgStatus s, const std::string &m) {
    err = m;
    return s;
  }
};

namespace {

#define HIPCHK(e, expr)                                                                   \
  do {                                                                                    \
    hipError_t _err = (expr);                                                             \
    if (_err != hipSuccess) {                                                             \
      return (e)->fail(TRG_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_err)); \
    }                                                                                     \
  } while (0)

template <typename T>
hipError_t alloc_pinned(PinnedBuf<T> &b, size_t cap) {
  if (b.cap >= cap) return hipSuccess;
  if (b.h) (void)hipHostFree(b.h);
  if (b.d) (void)hipFree(b.d);
  b.h = nullptr;
  b.d = nullptr;
  b.cap = 0;
  hipError_t e = hipHostMalloc((void **)&b.h, cap * sizeof(T), hipHostMallocDefault);
  if (e != hipSuccess) return e;
  e = hipMalloc((void **)&b.d, cap * sizeof(T));
  if (e != hipSuccess) return e;
  b.cap = cap;
  return hipSuccess;
}
template <typename T>
void free_pinned(PinnedBuf<T> &b) {
  if (b.h) (void)hipHostFree(b.h);
  if (b.d) (void)hipFree(b.d);
  b.h = nullptr;
  b.d = nullptr;
  b.cap = 0;
}

QueryParams qparams(const TrgEngine *e) {
  QueryParams q;
  q.robot_size = e->prm.robot_size;
  q.height_threshold = e->prm.height_threshold;
  q.collision_threshold = e->prm.collision_threshold;
  q.expand_dist = e->prm.expand_dist;
  q.sample_num = e->prm.sample_num;
  q.core_x0 = e->core[0];
  q.core_y0 = e->core[1];
  q.core_x1 = e->core[2];
  q.core_y1 = e->core[3];
  q.gate_margin = e->gate_margin;
  return q;
}

void free_map(DevMap &m) {
  if (m.x) (void)hipFree(m.x);
  if (m.y) (void)hipFree(m.y);
  if (m.z) (void)hipFree(m.z);
  if (m.perm) (void)hipFree(m.perm);
  if (m.pt) (void)hipFree(m.pt);
  if (m.cell_start) (void)hipFree(m.cell_start);
  m.top_wait();
  m = DevMap();
}

inline float key_to_float(unsigned k) {
  unsigned b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  float f;
  memcpy(&f, &b, 4);
  return f;
}

#include "trg_engine_map.ipp"

// ---- sampler table -----------------------------------------------------------------------------
TrgStatus ensure_sampler(TrgEngine *e, const TrgSampler *smp) {
  TrgSampler want = smp ? *smp : e->sampler;
  if (want.table_bits < 2 || want.table_bits > 20) want.table_bits = 16;
  e->sampler = want;
  if (e->table_bits_dev == want.table_bits && e->d_cos) return TRG_OK;
  const size_t n = (size_t)1 << want.table_bits;
  e->cos_t.resize(n);
  e->sin_t.resize(n);
  for (size_t k = 0; k < n; ++k) {
    // the reference's `float angle = distr_(gen_) * 2 * M_PI; cos(angle), sin(angle)` (trg.cpp:395-397)
    float u = (float)k / (float)n;
    float angle = u * 2 * M_PI;
    e->cos_t[k] = cos(angle);
    e->sin_t[k] = sin(angle);
  }
  if (e->d_cos) (void)hipFree(e->d_cos);
  if (e->d_sin) (void)hipFree(e->d_sin);
  e->d_cos = e->d_sin = nullptr;
  HIPCHK(e, hipMalloc((void **)&e->d_cos, n * sizeof(float)));
  HIPCHK(e, hipMalloc((void **)&e->d_sin, n * sizeof(float)));
  HIPCHK(e, hipMemcpy(e->d_cos, e->cos_t.data(), n * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(e->d_sin, e->sin_t.data(), n * sizeof(float), hipMemcpyHostToDevice));
  e->table_bits_dev = want.table_bits;
  return TRG_OK;
}

inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
inline uint32_t sample_hash(uint32_t seed, uint32_t epoch, uint32_t id, uint32_t trial) {
  uint32_t h = fmix32(seed ^ 0x9E3779B9u);
  h = fmix32(h + epoch * 0x9E3779B9u + 0x7F4A7C15u);
  h = fmix32(h + id * 0x85EBCA6Bu + 0x165667B1u);
  h = fmix32(h + trial * 0xC2B2AE35u + 0x27D4EB2Fu);
  return h;
}
inline float sampler_uniform(const TrgEngine *e, uint32_t k) {
  uint32_t h = sample_hash(e->sampler.seed, e->epoch, 0xFFFFFFFFu, k);
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

#include "trg_engine_probe.ipp"

// the stitched rows belong to the graph they were assembled from: every change of the global graph voids them
void invalidate_stitched(TrgEngine *e) {
  e->stitched_on_device = false;
  e->stitched_edges = 0;
  e->csr_stitched.clear();
}

// ---- graph state helpers -----------------------------------------------------------------------
void reset_graph_global(TrgEngine *e) {
  e->dev_csr_valid = false;
  e->nx.clear();
  e->ny.clear();
  e->nz.clear();
  e->nstate.clear();
  e->ncid.clear();
  e->order_map.clear();
  e->edges.reset(0);
  e->node_id = 0;
  e->kd.clear();
  e->kd_valid = true;  // empty tree is trivially in sync
  e->kd_order_dirty = false;
  e->kd_insert_order.clear();
  e->goal_node = -1;
  e->graph_version++;
  invalidate_stitched(e);
}

// the node map's keys in iteration order, whichever representation is current
void node_map_order(const TrgEngine *e, std::vector<int> &out) {
  if (e->real_map_stale) {
    e->nodes_sim.iteration_order(out);
  } else {
    out.clear();
    out.reserve(e->order_map.size());
    for (auto &kv : e->order_map) out.push_back(kv.first);
  }
}
// rebuild the real container exactly as cleanGraph left it: new_nodes[new_id] for the dense new ids,
// then nodes = new_nodes (trg.cpp:502, 526)
void ensure_real_map(TrgEngine *e) {
  if (!e->real_map_stale) return;
  std::unordered_map<int, int> fresh;
  const int n = (int)e->nodes_sim.size();
  for (int k = 0; k < n; ++k) fresh[k] = k;
  e->order_map = std::move(fresh);
  e->real_map_stale = false;
}

void grid_rebuild(TrgEngine *e) {
  const DevMap &m = e->gmap;
  float x0 = m.bounds[0], y0 = m.bounds[1], x1 = m.bounds[2], y1 = m.bounds[3];
  for (size_t i = 0; i < e->nx.size(); ++i) {
    x0 = std::min(x0, e->nx[i]);
    x1 = std::max(x1, e->nx[i]);
    y0 = std::min(y0, e->ny[i]);
    y1 = std::max(y1, e->ny[i]);
  }
  const float pad = e->prm.expand_dist * 2 + e->prm.robot_size;
  float cell = e->prm.robot_size > 0 ? e->prm.robot_size : 0.3f;
  while (((double)(x1 - x0 + 2 * pad) / cell + 5) * ((double)(y1 - y0 + 2 * pad) / cell + 5) > 128e6)
    cell *= 2;
  e->grid.reset(x0 - pad, y0 - pad, x1 + pad, y1 + pad, cell);
  for (size_t i = 0; i < e->nx.size(); ++i) e->grid.insert(e->nx[i], e->ny[i]);
}

// bring the reference-shaped kd replica in sync with the node set (lazy: the replay only needs it
// for ties and for step 3; queries need it for hit order)
// after a device build the node-tree refill order (the node map's iteration order, trg.cpp:528-530)
// is derived on demand
void materialize_kd_order(TrgEngine *e) {
  if (!e->kd_order_dirty) return;
  node_map_order(e, e->kd_insert_order);
  e->kd_order_dirty = false;
  e->kd_valid = false;
}

void kd_sync(TrgEngine *e) {
  materialize_kd_order(e);
  if (!e->kd_valid) {
    e->kd.clear();
    e->kd_valid = true;
  }
  while (e->kd.size() < e->kd_insert_order.size()) {
    const int id = e->kd_insert_order[e->kd.size()];
    e->kd.insert(e->nx[id], e->ny[id], id);
  }
}

int add_node_host(TrgEngine *e, float x, float y, float z, int state) {
  const int id = e->node_id;
  e->nx.push_back(x);
  e->ny.push_back(y);
  e->nz.push_back(z);
  e->nstate.push_back(state);
  e->ncid.push_back((int)e->ncid.size());
  // graph.nodes[node_id] = node (trg.cpp:248): into whichever representation of the container is current
  // (after a device build or a cleanGraph the keys are dense and ascending: the O(1) replica serves)
  if (e->real_map_stale) {
    e->nodes_sim.insert_next();
  } else {
    e->order_map[id] = id;
  }
  e->kd_insert_order.push_back(id);
  e->grid.insert(x, y);
  e->edges.grow_nodes(e->nx.size());
  e->node_id++;
  e->stats.created_nodes++;
  return id;
}

// nearest existing node exactly as kd_nearest2 on node_tree would answer (trg.cpp:408-409)
int nearest_node(TrgEngine *e, float qx, float qy) {
  bool tie = false;
  int s = e->grid.nearest(qx, qy, &tie);
  if (tie) {
    e->stats.nn_ties++;
    kd_sync(e);
    s = e->kd.nearest(qx, qy);
  }
  return s;
}

#include "trg_engine_replay.ipp"

#include "trg_engine_clean.ipp"

}  // namespace

#include "trg_engine_bfs.inc"

namespace {
// host-side edge pool / node grid rebuilt from the CSR after a device build (lazy)
void ensure_pool(TrgEngine *e) {
  if (e->pool_valid) return;
  const Csr &g = e->csr_global;
  const size_t V = g.state.size();
  std::vector<int> offs(g.rowptr.data(), g.rowptr.data() + V + 1);
  e->edges.reset(0);
  e->edges.alloc_rows(offs);
  parallel_ranges(V, [&](size_t i0, size_t i1) {
    const size_t a = (size_t)offs[i0], b = (size_t)offs[i1];
    if (b > a) {
      memcpy(e->edges.dst.data() + a, g.col.data() + a, (b - a) * sizeof(int));
      memcpy(e->edges.w.data() + a, g.w.data() + a, (b - a) * sizeof(float));
      memcpy(e->edges.dist.data() + a, g.dist.data() + a, (b - a) * sizeof(float));
    }
    e->edges.link_rows(offs, i0, i1);
  });
  e->pool_valid = true;
}
void ensure_host_grid(TrgEngine *e) {
  if (e->host_grid_valid) return;
  grid_rebuild(e);
  e->host_grid_valid = true;
}
}  // namespace

// =================================== C ABI ======================================================
extern "C" {

TrgStatus trg_engine_create(const TrgParams *params, int device, TrgEngine **out) {
  if (!params || !out) return TRG_ERR_INVALID_ARG;
  *out = nullptr;
  TrgEngine *e = new TrgEngine();
  e->prm = *params;
  e->device = device;
  *out = e;  // handed out even on failure so the caller can read last_error
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    return e->fail(TRG_ERR_DEVICE, "no HIP device visible: the TRG engine has no CPU fallback");
  }
  if (device < 0 || device >= ndev) return e->fail(TRG_ERR_INVALID_ARG, "bad device ordinal");
  HIPCHK(e, hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(e, hipGetDeviceProperties(&prop, device));
  e->arch = prop.gcnArchName;
  if (e->arch.rfind("gfx950", 0) != 0) {
    return e->fail(TRG_ERR_DEVICE, "kernels are built for gfx950 only, device is " + e->arch);
  }
  {
    int pr_least = 0, pr_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
    if (getenv("TRG_MAIN_PRIO") && atoi(getenv("TRG_MAIN_PRIO")) != 0)  // (measurements) the level loop at the highest priority
      HIPCHK(e, hipStreamCreateWithPriority(&e->s_main, hipStreamNonBlocking, pr_greatest));
    else
      HIPCHK(e, hipStreamCreateWithFlags(&e->s_main, hipStreamNonBlocking));
    // TRG_EDGE_CU_MASK=n: the second stream is confined to n compute units (experiment)
    const char *cm = getenv("TRG_EDGE_CU_MASK");
    int ncu_edge = cm ? atoi(cm) : 0;
    if (ncu_edge > 0) {
      uint32_t mask[8] = {0};
      const int stride = 256 / std::max(1, std::min(ncu_edge, 256));
      for (int k = 0; k < ncu_edge && k * stride < 256; ++k) mask[(k * stride) / 32] |= 1u << ((k * stride) % 32);
      HIPCHK(e, hipExtStreamCreateWithCUMask(&e->s_edge, 8, mask));
    } else {
      // the priority of the main stream (default; lowest leaves a longer tail after the loop: measured
      // 0.5 ms slower); TRG_EDGE_PRIO=1 (measurements): lowest
      const bool low = getenv("TRG_EDGE_PRIO") && atoi(getenv("TRG_EDGE_PRIO")) != 0;
      HIPCHK(e, hipStreamCreateWithPriority(&e->s_edge, hipStreamNonBlocking, low ? pr_least : 0));
    }
    HIPCHK(e, hipStreamCreateWithFlags(&e->s_aux, hipStreamNonBlocking));
  }
  HIPCHK(e, hipMalloc((void **)&e->d_ctr, COUNTER_SHARDS * sizeof(DeviceCounters)));
  HIPCHK(e, hipMemset(e->d_ctr, 0, COUNTER_SHARDS * sizeof(DeviceCounters)));
  HIPCHK(e, hipMalloc((void **)&e->d_bounds, 4 * sizeof(unsigned)));
  // step 3 of expandGraph is compiled in or out by this fp comparison (trg.cpp:429)
  e->step3 = (e->prm.expand_dist - e->prm.robot_size) < 0.25 * e->prm.expand_dist;
  e->device_ok = true;
  e->bfs = new BfsBuffers();
  e->stitch = new StitchBufs();
  e->uploader = new Uploader();
  if (const char *env = getenv("TRG_REPLAY")) e->use_device_bfs = std::string(env) != "host";
  if (const char *env = getenv("TRG_PRESAMPLE")) e->presample = atoi(env) != 0;            // (A/B measurements)
  if (const char *env = getenv("TRG_RESOLVE_TICKETS")) e->resolve_tickets = atoi(env);
  reset_graph_global(e);
  return TRG_OK;
}

void trg_engine_destroy(TrgEngine *e) {
  if (!e) return;
  if (e->device_ok) {
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    free_map(e->gmap);
    free_map(e->lmap);
    Chunk *all_chunks[TrgEngine::NCHUNK + 1];
    for (int i = 0; i < TrgEngine::NCHUNK; ++i) all_chunks[i] = &e->chunks[i];
    all_chunks[TrgEngine::NCHUNK] = &e->root_chunk;
    for (Chunk *cp : all_chunks) {
      Chunk &c = *cp;
      if (c.done) (void)hipEventDestroy(c.done);
      if (c.t0) (void)hipEventDestroy(c.t0);
      if (c.t1) (void)hipEventDestroy(c.t1);
      if (c.t2) (void)hipEventDestroy(c.t2);
      free_pinned(c.in_blob);
      free_pinned(c.out_blob);
      free_pinned(c.mt);
      if (c.d_mid) (void)hipFree(c.d_mid);
    }
    for (EdgeBatch &b : e->ebatches) {
      if (b.done) (void)hipEventDestroy(b.done);
      if (b.t0) (void)hipEventDestroy(b.t0);
      if (b.t1) (void)hipEventDestroy(b.t1);
      free_pinned(b.p1);
      free_pinned(b.p2);
      free_pinned(b.weight);
      free_pinned(b.dist);
      free_pinned(b.status);
      if (b.d_mid) (void)hipFree(b.d_mid);
    }
    if (e->sy_mid) (void)hipFree(e->sy_mid);
    if (e->mt_set_d) (void)hipFree(e->mt_set_d);
    if (e->mt_walk_d) (void)hipFree(e->mt_walk_d);
    if (e->mt_set_h) (void)hipHostFree(e->mt_set_h);
    if (e->mt_walk_h) (void)hipHostFree(e->mt_walk_h);
    delete e->plan_scratch;
    if (e->uploader) {
      e->uploader->release();
      delete e->uploader;
    }
    if (e->bfs) {
      e->bfs->release();
      delete e->bfs;
    }
    exchange_release(e);
    if (e->stitch) {
      e->stitch->release();
      delete e->stitch;
    }
    free_pinned(e->sy_in);
    free_pinned(e->sy_in2);
    free_pinned(e->sy_f0);
    free_pinned(e->sy_f1);
    free_pinned(e->sy_i0);
    free_pinned(e->sy_i1);
    free_pinned(e->sy_i2);
    if (e->d_cos) (void)hipFree(e->d_cos);
    if (e->d_sin) (void)hipFree(e->d_sin);
    if (e->d_ctr) (void)hipFree(e->d_ctr);
    if (e->d_bounds) (void)hipFree(e->d_bounds);
    if (e->top_xy_d) (void)hipFree(e->top_xy_d);
    if (e->top_xy_h) (void)hipHostFree(e->top_xy_h);
    if (e->top_ev) (void)hipEventDestroy(e->top_ev);
    for (void *p : {(void *)e->idx_scratch.cell_of, (void *)e->idx_scratch.rank, (void *)e->idx_scratch.counts,
                    (void *)e->idx_scratch.tmp, e->idx_scratch.aos, (void *)e->idx_scratch.hist,
                    (void *)e->idx_scratch.base, (void *)e->idx_scratch.bin_tmp})
      if (p) (void)hipFree(p);
    if (e->s_main) (void)hipStreamDestroy(e->s_main);
    if (e->s_edge) (void)hipStreamDestroy(e->s_edge);
    if (e->s_aux) (void)hipStreamDestroy(e->s_aux);
  }
  delete e;
}

const char *trg_engine_last_error(const TrgEngine *e) { return e ? e->err.c_str() : "null engine"; }
const char *trg_engine_device_arch(const TrgEngine *e) { return e ? e->arch.c_str() : ""; }

#define REQUIRE_DEVICE(e)                                                              \
  do {                                                                                 \
    if (!(e)) return TRG_ERR_INVALID_ARG;                                              \
    if (!(e)->device_ok) return (e)->fail(TRG_ERR_DEVICE, "engine has no usable device"); \
    if (hipSetDevice((e)->device) != hipSuccess)                                       \
      return (e)->fail(TRG_ERR_DEVICE, "hipSetDevice failed");                         \
  } while (0)

TrgStatus trg_engine_set_global_map(TrgEngine *e, const float *xyz, size_t n, size_t stride) {
  REQUIRE_DEVICE(e);
  if ((n && !xyz) || stride < 3) return e->fail(TRG_ERR_INVALID_ARG, "bad map arguments");
  auto t0 = Clock::now();
  TrgStatus st = upload_and_build(e, e->gmap, xyz, n, stride);
  e->stats.ms_set_map_total = ms_since(t0);
  return st;
}

TrgStatus trg_engine_set_global_map_device(TrgEngine *e, const float *d_xyz, size_t n,
                                           size_t stride) {
  REQUIRE_DEVICE(e);
  if ((n && !d_xyz) || stride < 3) return e->fail(TRG_ERR_INVALID_ARG, "bad map arguments");
  if (n == 0) {
    e->gmap.n = 0;
    e->gmap.valid = false;
    return TRG_OK;
  }
  return build_map(e, e->gmap, d_xyz, n, stride);
}

TrgStatus trg_engine_set_local_map(TrgEngine *e, const float start_xy[2], const float *xyz,
                                   size_t n, size_t stride) {
  REQUIRE_DEVICE(e);
  if (!start_xy || (n && !xyz) || stride < 3) return e->fail(TRG_ERR_INVALID_ARG, "bad arguments");
  e->local_root[0] = start_xy[0];
  e->local_root[1] = start_xy[1];
  TrgStatus st = upload_and_build(e, e->lmap, xyz, n, stride);
  if (st != TRG_OK) return st;
  return set_local_graph(e);
}

TrgStatus trg_engine_reset_map(TrgEngine *e, TrgKind kind) {
  REQUIRE_DEVICE(e);
  DevMap *m = pick_map(e, kind);
  m->top_wait();  // (a helper preparing the map's tie-break structures still reads it)
  m->n = 0;
  m->valid = false;
  return TRG_OK;
}

TrgStatus trg_engine_reset_graph(TrgEngine *e, TrgKind kind) {
  REQUIRE_DEVICE(e);
  if (kind == TRG_KIND_LOCAL) {
    e->local_nodes.clear();
    e->local_map.clear();
    e->lkd.clear();
  } else {
    ensure_real_map(e);
    reset_graph_global(e);
    e->csr_global.clear();
  }
  return TRG_OK;
}

TrgStatus trg_engine_init_graph(TrgEngine *e, const float start_xyz[3], const TrgSampler *sampler) {
  REQUIRE_DEVICE(e);
  if (!start_xyz) return e->fail(TRG_ERR_INVALID_ARG, "null start");
  if (!e->gmap.valid || e->gmap.n == 0) return e->fail(TRG_ERR_NO_MAP, "Map is empty");
  auto t_total = Clock::now();
  TrgStatus st = ensure_sampler(e, sampler);
  if (st != TRG_OK) return st;
  // per-build stats (map-index figures are kept)
  {
    TrgStats keep = e->stats;
    e->stats = TrgStats();
    e->stats.map_points = keep.map_points;
    e->stats.ms_index_build = keep.ms_index_build;
    e->stats.bytes_index_build = keep.bytes_index_build;
    e->stats.ms_set_map_total = keep.ms_set_map_total;
  }
  // (in the main stream: it is non-blocking, a plain memset would not be ordered with the kernels)
  HIPCHK(e, hipMemsetAsync(e->d_ctr, 0, COUNTER_SHARDS * sizeof(DeviceCounters), e->s_main));
  e->lv_hits_sample = e->lv_hits_spec = 0;
  const bool want_device = e->use_device_bfs && (!e->step3 || e->step3_device);
  if (!want_device) ensure_real_map(e);
  MapOrderSim sim_before;  // container history as of before this build (for the fallback)
  if (want_device) {
    if (!e->real_map_stale) e->nodes_sim.adopt_bucket_state(e->order_map);
    sim_before = e->nodes_sim;
  }
  const bool stale_before = e->real_map_stale;
  reset_graph_global(e);
  e->real_map_stale = stale_before;
  e->pool_valid = true;
  e->epoch = e->epoch_base;
  e->calls.clear();
  e->pending_calls.clear();
  e->csr_pre.clear();

  // root seeding, trg.cpp:44-56
  e->root_pos[0] = start_xyz[0];
  e->root_pos[1] = start_xyz[1];
  float rx = e->root_pos[0], ry = e->root_pos[1], rz = 0.0f;
  rx = rx + e->prm.expand_dist;
  int cnt = 0;
  for (;;) {
    float xy[2] = {rx, ry};
    int32_t flag = 1;
    st = collision_sync(e, e->gmap, e->prm.collision_threshold, xy, 1, &flag, nullptr, nullptr);
    if (st != TRG_OK) return st;
    if (!(rx >= e->core[0] && rx < e->core[2] && ry >= e->core[1] && ry < e->core[3])) flag = 1;
    if (!flag) {
      int32_t found = 0;
      st = nearest_z_sync(e, e->gmap, xy, 1, &rz, &found);
      if (st != TRG_OK) return st;
      break;
    }
    if (cnt > 100) return e->fail(TRG_ERR_NO_ROOT, "Failed to generate root node");
    const float u0 = sampler_uniform(e, 2 * cnt), u1 = sampler_uniform(e, 2 * cnt + 1);
    rx = rx + e->prm.expand_dist * u0;
    ry = ry + e->prm.expand_dist * u1;
    cnt++;
  }

  if (want_device) {
    st = build_graph_device(e, rx, ry, rz);
    if (st == TRG_OK) {
      read_counters(e);
      e->stats.used_device_bfs = 1;
      e->stats.ms_init_graph_total = ms_since(t_total);
      return TRG_OK;
    }
    if (e->bfs_fallback_reason.empty()) return st;
    // the device path declined (capacity, or an exact fp32 tie whose winner depends on the
    // reference kd-tree's shape): redo the build with the host replay, which handles those
    e->stats.bfs_fallbacks++;
    // (kernels of the abandoned attempt may still be in flight: the streams do not synchronise with
    // plain copies / memsets)
    HIPCHK(e, hipStreamSynchronize(e->s_main));
    HIPCHK(e, hipStreamSynchronize(e->s_edge));
    HIPCHK(e, hipMemset(e->d_ctr, 0, COUNTER_SHARDS * sizeof(DeviceCounters)));
    e->lv_hits_sample = e->lv_hits_spec = 0;
    e->nodes_sim = sim_before;
    e->real_map_stale = stale_before;
    ensure_real_map(e);
    reset_graph_global(e);
  }

  grid_rebuild(e);
  e->host_grid_valid = true;
  add_node_host(e, rx, ry, rz, TRG_NODE_VALID);
  st = expand_bfs(e, e->node_id - 1);
  if (st != TRG_OK) return st;
  auto t_fin = Clock::now();
  st = flush_pending(e, true);
  if (st != TRG_OK) return st;
  apply_calls(e, 0);
  if (e->keep_preclean) snapshot_csr(e, e->csr_pre);
  clean_graph(e);
  snapshot_csr(e, e->csr_global);
  e->graph_version++;
  invalidate_stitched(e);
  e->stats.ms_finalize_host = ms_since(t_fin);
  read_counters(e);
  e->stats.ms_init_graph_total = ms_since(t_total);
  return TRG_OK;
}

TrgStatus trg_engine_set_option(TrgEngine *e, const char *key, const char *value) {
  if (!e || !key || !value) return TRG_ERR_INVALID_ARG;
  const std::string k(key), v(value);
  if (k == "replay") {
    if (v == "host") e->use_device_bfs = false;
    else if (v == "device") e->use_device_bfs = true;
    else return e->fail(TRG_ERR_INVALID_ARG, "replay must be host or device");
    return TRG_OK;
  }
  if (k == "debug_gate_margin") {
    e->gate_margin = (float)atof(v.c_str());
    return TRG_OK;
  }
  if (k == "debug_tie_every") {
    e->debug_tie_every = atoi(v.c_str());
    return TRG_OK;
  }
  if (k == "debug_spec_bound") {
    e->debug_spec_bound = atoi(v.c_str());
    return TRG_OK;
  }
  if (k == "defer_overlap") {
    e->defer_overlap = atoi(v.c_str());
    return TRG_OK;
  }
  if (k == "debug_stall_level") {
    e->debug_stall_level = atoi(v.c_str());
    return TRG_OK;
  }
  if (k == "debug_lookback_level") {
    e->debug_lookback_level = atoi(v.c_str());
    return TRG_OK;
  }
  if (k == "step3_device") {
    e->step3_device = v != "0";
    return TRG_OK;
  }
  if (k == "debug_call_stride") {
    e->debug_call_stride = v != "0";
    return TRG_OK;
  }
  if (k == "debug_wait_rerun") {
    e->debug_wait_rerun = v != "0";
    return TRG_OK;
  }
  if (k == "presample") {
    e->presample = v != "0";
    return TRG_OK;
  }
  if (k == "resolve_tickets") {
    e->resolve_tickets = atoi(v.c_str());
    return TRG_OK;
  }
  if (k == "tie_inplace") {
    e->tie_inplace = v != "0";
    return TRG_OK;
  }
  if (k == "debug_fallback_level") {
    e->debug_fallback_level = atoi(v.c_str());
    return TRG_OK;
  }
  if (k == "keep_preclean") {
    e->keep_preclean = v != "0";
    return TRG_OK;
  }
  return e->fail(TRG_ERR_INVALID_ARG, "unknown option " + k);
}

TrgStatus trg_engine_set_tile(TrgEngine *e, const float core_xyxy[4], uint32_t epoch) {
  if (!e) return TRG_ERR_INVALID_ARG;
  if (core_xyxy) {
    for (int i = 0; i < 4; ++i) e->core[i] = core_xyxy[i];
  } else {
    e->core[0] = e->core[1] = -INFINITY;
    e->core[2] = e->core[3] = INFINITY;
  }
  e->epoch_base = epoch;
  return TRG_OK;
}

const char *trg_engine_fallback_reason(const TrgEngine *e) {
  return e ? e->bfs_fallback_reason.c_str() : "";
}

TrgStatus trg_engine_update_graph(TrgEngine *e) {
  REQUIRE_DEVICE(e);
  if (!e->gmap.valid) return e->fail(TRG_ERR_NO_MAP, "Map is empty");
  TrgStatus st = ensure_sampler(e, nullptr);
  if (st != TRG_OK) return st;
  e->epoch++;
  e->dev_csr_valid = false;
  const bool trace_up = getenv("TRG_TIMING") != nullptr;
  auto t_up = Clock::now();
  auto lap_up = [&](const char *what) {
    if (trace_up) fprintf(stderr, "[trg update] %-28s %8.3f ms\n", what, ms_since(t_up));
  };
  ensure_pool(e);
  ensure_host_grid(e);
  materialize_kd_order(e);  // nodes created below are appended to the existing insertion order
  e->calls.clear();
  e->pending_calls.clear();
  lap_up("host structures ready");

  // per local node: invalidate / keep frontier / revalidate (trg.cpp:464-481)
  const size_t L = e->local_nodes.size();
  std::vector<float> xy(2 * L);
  for (size_t i = 0; i < L; ++i) {
    xy[2 * i] = e->nx[e->local_nodes[i]];
    xy[2 * i + 1] = e->ny[e->local_nodes[i]];
  }
  std::vector<int32_t> col(L, 0), fro(L, 0);
  if (L) {
    st = collision_sync(e, e->lmap, e->prm.update_collision_threshold, xy.data(), L, col.data(),
                        nullptr, nullptr);
    if (st != TRG_OK) return st;
    st = trg_engine_is_frontier_batch(e, xy.data(), L, fro.data());
    if (st != TRG_OK) return st;
  }
  std::vector<int> expand_queue;
  for (size_t i = 0; i < L; ++i) {
    const int id = e->local_nodes[i];
    const float px = e->nx[id], py = e->ny[id];
    if (norm2f(px - e->local_root[0], py - e->local_root[1]) > 2.0 * e->prm.expand_dist) {
      if (col[i] || e->edges.deg[id] < 1) {
        e->nstate[id] = TRG_NODE_INVALID;
        continue;
      }
    }
    if (fro[i] && e->nstate[id] == TRG_NODE_FRONTIER) {
      e->nstate[id] = TRG_NODE_FRONTIER;
      expand_queue.push_back(id);
      continue;
    }
    expand_queue.push_back(id);
    e->nstate[id] = TRG_NODE_VALID;
  }
  // isFrontier looked at the node set as it stood; expansions below mutate it, and the reference
  // evaluates isFrontier lazily inside the same loop BEFORE any expansion, so this order is exact.
  // The GPU results of a root (its samples and their speculative parent edges) depend on the map
  // and on (seed, epoch, id) only, so they are fetched for all roots in bulk; the expansions
  // themselves are replayed one root after the other, as the reference runs them.  wireEdge's
  // dedupe does not influence any node decision, so the logged calls are evaluated and applied once
  // at the end, in program order (first successful call per pair wins).
  lap_up("local nodes classified");
  st = ensure_chunks(e);
  if (st != TRG_OK) return st;
  for (size_t g0 = 0; g0 < expand_queue.size(); g0 += TrgEngine::CHUNK_MAX) {
    const int cnt = (int)std::min<size_t>(TrgEngine::CHUNK_MAX, expand_queue.size() - g0);
    st = submit_nodes(e, e->root_chunk, expand_queue.data() + g0, cnt);
    if (st != TRG_OK) return st;
    st = wait_chunk(e, e->root_chunk);
    if (st != TRG_OK) return st;
    for (int i = 0; i < cnt; ++i) {
      st = expand_bfs(e, expand_queue[g0 + i], &e->root_chunk, i);
      if (st != TRG_OK) return st;
    }
  }
  lap_up("roots expanded");
  st = flush_pending(e, true);
  if (st != TRG_OK) return st;
  lap_up("deferred edges evaluated");
  std::vector<int32_t> member_old;  // local-graph membership of the nodes as they are now (positions do not change)
  st = local_membership(e, member_old);
  if (st != TRG_OK) return st;
  apply_calls(e, 0);
  lap_up("calls applied");
  if (e->keep_preclean) snapshot_csr(e, e->csr_pre);
  clean_graph(e);
  lap_up("cleanGraph");
  snapshot_csr(e, e->csr_global);
  e->graph_version++;
  invalidate_stitched(e);
  lap_up("CSR snapshot");
  read_counters(e);
  std::vector<int32_t> member_new(e->nx.size(), 0);
  for (size_t k = 0; k < member_new.size() && k < e->last_new2old.size(); ++k)
    member_new[k] = member_old[e->last_new2old[k]];
  st = set_local_graph(e, &member_new);
  lap_up("setLocalGraph");
  return st;  // cleanGraph(true) -> setLocalGraph (trg.cpp:532-534)
}

TrgStatus trg_engine_export_csr(TrgEngine *e, TrgKind kind, TrgCsrView *out) {
  if (!e || !out) return TRG_ERR_INVALID_ARG;
  Csr *c = nullptr;
  if (kind == TRG_KIND_GLOBAL) {
    c = &e->csr_global;
    if (e->pool_valid && (c->rowptr.empty() || c->state.size() != e->nx.size())) snapshot_csr(e, *c);
  } else if (kind == TRG_KIND_PRECLEAN) {
    c = &e->csr_pre;
  } else if (kind == TRG_KIND_STITCHED) {
    c = &e->csr_stitched;
    const TrgStatus fs = stitch_fetch(e);
    if (fs != TRG_OK) return fs;
  } else {
    // local graph: the global rows of the local member nodes
    c = &e->csr_local;
    ensure_pool(e);
    Csr full;
    snapshot_csr(e, full);
    c->clear();
    c->rowptr.push_back(0);
    for (int id : e->local_nodes) {
      c->xyz.append(full.xyz.begin() + 3 * id, full.xyz.begin() + 3 * id + 3);
      c->state.push_back(full.state[id]);
      c->cid.push_back(id);  // for the local view: the node's global id
      for (int k = full.rowptr[id]; k < full.rowptr[id + 1]; ++k) {
        c->col.push_back(full.col[k]);
        c->w.push_back(full.w[k]);
        c->dist.push_back(full.dist[k]);
      }
      c->rowptr.push_back((int)c->col.size());
    }
  }
  if (c->rowptr.empty()) c->rowptr.push_back(0);
  out->num_nodes = (int32_t)c->state.size();
  out->num_edges = (int32_t)c->col.size();
  out->node_xyz = c->xyz.data();
  out->node_state = c->state.data();
  out->rowptr = c->rowptr.data();
  out->col = c->col.data();
  out->weight = c->w.data();
  out->dist = c->dist.data();
  out->creation_id = c->cid.data();
  return TRG_OK;
}

// ---- JSON persistence (schema of trg.cpp:130-177) ---------------------------------------------
TrgStatus trg_engine_save_json(TrgEngine *e, const char *path) {
  if (!e || !path) return TRG_ERR_INVALID_ARG;
  std::string p(path);
  // save_path.extension().empty() -> append ".json" (trg.cpp:135-137)
  {
    size_t slash = p.find_last_of('/');
    size_t dot = p.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) p += ".json";
  }
  std::ofstream f(p);
  if (!f) return e->fail(TRG_ERR_IO, "cannot open " + p);
  ensure_pool(e);
  // nodes/edges are listed in the node map's iteration order, like the reference
  std::vector<int> map_order;
  node_map_order(e, map_order);
  GraphJson g;
  g.nodes.reserve(map_order.size());
  for (int id : map_order) {
    g.nodes.push_back({id, {e->nx[id], e->ny[id], e->nz[id]}, e->nstate[id]});
    for (int ed = e->edges.head[id]; ed >= 0; ed = e->edges.next[ed])
      g.edges.push_back({id, e->edges.dst[ed], e->edges.w[ed], e->edges.dist[ed]});
  }
  write_graph_json(f, g);
  f.close();
  if (!f) return e->fail(TRG_ERR_IO, "write failed: " + p);
  return TRG_OK;
}

TrgStatus trg_engine_load_json(TrgEngine *e, const char *path) {
  REQUIRE_DEVICE(e);
  if (!path) return TRG_ERR_INVALID_ARG;
  std::ifstream f(path);
  if (!f) return e->fail(TRG_ERR_IO, std::string("File not found: ") + path);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string txt = ss.str();
  GraphJson gj;
  std::string perr;
  if (!parse_graph_json(txt, gj, perr) || !validate_graph_json(gj, perr)) return e->fail(TRG_ERR_IO, perr);
  const std::vector<GraphJson::N> &nodes = gj.nodes;
  const std::vector<GraphJson::Ed> &eds = gj.edges;
  const size_t V = nodes.size();
  ensure_real_map(e);
  reset_graph_global(e);
  e->nx.assign(V, 0);
  e->ny.assign(V, 0);
  e->nz.assign(V, 0);
  e->nstate.assign(V, 0);
  e->ncid.assign(V, 0);
  e->edges.reset(V);
  for (auto &n : nodes) {
    e->nx[n.id] = n.p[0];
    e->ny[n.id] = n.p[1];
    e->nz[n.id] = n.p[2];
    e->nstate[n.id] = n.state;
    e->ncid[n.id] = n.id;
    e->order_map[n.id] = n.id;          // global_graph.nodes[id] = node_ptr, file order
    e->kd_insert_order.push_back(n.id);  // kd_insert2 in file order (trg.cpp:103)
  }
  e->node_id = (int)V;
  for (auto &ed : eds) e->edges.push(ed.s, ed.t, ed.w, ed.d);
  e->kd_valid = false;
  grid_rebuild(e);
  e->host_grid_valid = true;
  e->pool_valid = true;
  snapshot_csr(e, e->csr_global);
  e->graph_version++;
  invalidate_stitched(e);
  return TRG_OK;
}

// ---- planning (host A*, trg.cpp:537-565, 603-690): trg_engine_plan.ipp ----------------------------
}  // extern "C"

#include "trg_engine_plan.ipp"

extern "C" {

// ---- probes --------------------------------------------------------------------------------------
TrgStatus trg_engine_is_collision_batch(TrgEngine *e, TrgKind map, float threshold, const float *xy,
                                        size_t m, int32_t *flag, int32_t *cnt, int32_t *n) {
  REQUIRE_DEVICE(e);
  if (m && !xy) return e->fail(TRG_ERR_INVALID_ARG, "null positions");
  return collision_sync(e, *pick_map(e, map), threshold, xy, m, flag, cnt, n);
}

TrgStatus trg_engine_nearest_z_batch(TrgEngine *e, TrgKind map, const float *xy, size_t m, float *z) {
  REQUIRE_DEVICE(e);
  if (m && (!xy || !z)) return e->fail(TRG_ERR_INVALID_ARG, "null arguments");
  return nearest_z_sync(e, *pick_map(e, map), xy, m, z, nullptr);
}

TrgStatus trg_engine_edge_risk_batch(TrgEngine *e, TrgKind map, const float *p1_xyz,
                                     const float *p2_xyz, size_t m, int32_t *status, int32_t *n_pts,
                                     float *weight, float *dist) {
  REQUIRE_DEVICE(e);
  if (m && (!p1_xyz || !p2_xyz)) return e->fail(TRG_ERR_INVALID_ARG, "null arguments");
  return edges_sync(e, *pick_map(e, map), p1_xyz, p2_xyz, m, status, n_pts, weight, dist, true);
}

TrgStatus trg_engine_voxel_filter(TrgEngine *e, const float *xyz, size_t n, size_t stride, float leaf,
                                  float *out_xyz, size_t *n_out, int32_t *passthrough) {
  REQUIRE_DEVICE(e);
  if (!n_out || (n && (!xyz || !out_xyz))) return e->fail(TRG_ERR_INVALID_ARG, "null arguments");
  if (stride < 3) return e->fail(TRG_ERR_INVALID_ARG, "stride must be >= 3 floats");
  if (!(leaf > 0.0f)) return e->fail(TRG_ERR_INVALID_ARG, "leaf size must be positive");
  *n_out = 0;
  if (passthrough) *passthrough = 0;
  if (n == 0) return TRG_OK;
  float *d_in = nullptr, *d_out = nullptr;
  HIPCHK(e, hipMalloc((void **)&d_in, n * stride * sizeof(float)));
  hipError_t he = hipMalloc((void **)&d_out, n * 3 * sizeof(float));
  if (he == hipSuccess)
    he = hipMemcpyAsync(d_in, xyz, n * stride * sizeof(float), hipMemcpyHostToDevice, e->s_main);
  int status = 0;
  size_t m = 0;
  if (he == hipSuccess) he = voxel_grid_filter(d_in, n, stride, leaf, d_out, &m, &status, e->s_main);
  if (he == hipSuccess && m)
    he = hipMemcpy(out_xyz, d_out, m * 3 * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (he != hipSuccess)
    return e->fail(TRG_ERR_DEVICE, std::string("voxel filter: ") + hipGetErrorString(he));
  *n_out = m;
  if (passthrough) *passthrough = status;
  return TRG_OK;
}

TrgStatus trg_engine_is_frontier_batch(TrgEngine *e, const float *xy, size_t m, int32_t *flag) {
  REQUIRE_DEVICE(e);
  if (m && (!xy || !flag)) return e->fail(TRG_ERR_INVALID_ARG, "null arguments");
  // trg.cpp:780-803: check = pos + 2*robot_size*normalize(pos - local root); frontier iff no global
  // node within robot_size of check AND no local-map point within robot_size/2 of it
  // "no global node within robot_size" only asks whether the range query is empty: the node grid answers
  // that without the tree replica (building it costs O(V log V) per update); the replica is consulted
  // only for a node within rounding of the radius, where the tree's pruning decides
  ensure_host_grid(e);
  std::vector<float> chk(2 * m);
  std::vector<int> hits;
  std::vector<char> blocked(m, 0);
  for (size_t i = 0; i < m; ++i) {
    float dx = xy[2 * i] - e->local_root[0], dy = xy[2 * i + 1] - e->local_root[1];
    const float sq = dx * dx + dy * dy;
    if (sq > 0.0f) {
      const float nrm = sqrtf(sq);
      dx = dx / nrm;
      dy = dy / nrm;
    }
    const float k = 2 * e->prm.robot_size;
    chk[2 * i] = xy[2 * i] + k * dx;
    chk[2 * i + 1] = xy[2 * i + 1] + k * dy;
    int w = e->grid.ready() ? e->grid.within(chk[2 * i], chk[2 * i + 1], e->prm.robot_size) : -1;
    if (w < 0) {
      kd_sync(e);
      e->kd.range(chk[2 * i], chk[2 * i + 1], e->prm.robot_size, hits);
      w = hits.empty() ? 0 : 1;
    }
    blocked[i] = w != 0;
  }
  std::vector<int32_t> n(m, 0);
  if (e->lmap.valid && m) {
    TrgStatus st = collision_sync(e, e->lmap, 0.0f, chk.data(), m, nullptr, nullptr, n.data(), nullptr,
                                  (float)(0.5 * e->prm.robot_size));
    if (st != TRG_OK) return st;
  }
  for (size_t i = 0; i < m; ++i) flag[i] = (!blocked[i] && n[i] == 0) ? 1 : 0;
  return TRG_OK;
}

TrgStatus trg_engine_get_stats(const TrgEngine *e, TrgStats *out) {
  if (!e || !out) return TRG_ERR_INVALID_ARG;
  *out = e->stats;
  return TRG_OK;
}

TrgStatus trg_engine_get_sampler_table(TrgEngine *e, float *cos_out, float *sin_out) {
  REQUIRE_DEVICE(e);
  TrgStatus st = ensure_sampler(e, nullptr);
  if (st != TRG_OK) return st;
  if (cos_out) memcpy(cos_out, e->cos_t.data(), e->cos_t.size() * sizeof(float));
  if (sin_out) memcpy(sin_out, e->sin_t.data(), e->sin_t.size() * sizeof(float));
  return TRG_OK;
}

TrgStatus trg_engine_debug_map_index(TrgEngine *e, TrgKind map, float *x, float *y, float *z,
                                     int32_t *perm, int32_t *grid_wh, float *origin_cell) {
  REQUIRE_DEVICE(e);
  DevMap &m = *pick_map(e, map);
  if (!m.valid) return e->fail(TRG_ERR_NO_MAP, "no map");
  if (x) HIPCHK(e, hipMemcpy(x, m.x, m.n * sizeof(float), hipMemcpyDeviceToHost));
  if (y) HIPCHK(e, hipMemcpy(y, m.y, m.n * sizeof(float), hipMemcpyDeviceToHost));
  if (z) HIPCHK(e, hipMemcpy(z, m.z, m.n * sizeof(float), hipMemcpyDeviceToHost));
  if (perm) HIPCHK(e, hipMemcpy(perm, m.perm, m.n * sizeof(int), hipMemcpyDeviceToHost));
  if (grid_wh) {
    grid_wh[0] = m.view.W;
    grid_wh[1] = m.view.H;
  }
  if (origin_cell) {
    origin_cell[0] = m.view.x0;
    origin_cell[1] = m.view.y0;
    origin_cell[2] = m.g;
  }
  return TRG_OK;
}

}  // extern "C"

#include "trg_engine_stitch.inc"
#include "trg_engine_exchange.inc"
