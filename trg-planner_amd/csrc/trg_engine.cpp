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

// ---- map index build ---------------------------------------------------------------------------
void start_map_top(TrgEngine *e, DevMap &m);

TrgStatus build_map(TrgEngine *e, DevMap &m, const float *d_xyz, size_t n, size_t stride) {
  m.top_wait();  // (a helper of the previous build still reads the arrays that are replaced below)
  auto t_host = Clock::now();
  m.valid = false;
  if (n == 0) {
    m.n = 0;
    return TRG_OK;
  }
  if (n > (size_t)0x7FFFFFF0) return e->fail(TRG_ERR_CAPACITY, "more than 2^31 map points");
  hipStream_t s = e->s_main;
  hipEvent_t ev0, ev1;
  HIPCHK(e, hipEventCreate(&ev0));
  HIPCHK(e, hipEventCreate(&ev1));
  HIPCHK(e, hipEventRecord(ev0, s));
  launch_init_bounds(e->d_bounds, s);
  launch_bounds(d_xyz, n, stride, e->d_bounds, s);
  unsigned hb[4];
  HIPCHK(e, hipMemcpyAsync(hb, e->d_bounds, sizeof(hb), hipMemcpyDeviceToHost, s));
  HIPCHK(e, hipStreamSynchronize(s));
  const float x0 = key_to_float(hb[0]), y0 = key_to_float(hb[1]);
  const float x1 = key_to_float(hb[2]), y1 = key_to_float(hb[3]);
  if (!(x1 >= x0) || !(y1 >= y0) || !std::isfinite(x0) || !std::isfinite(x1) ||
      !std::isfinite(y0) || !std::isfinite(y1)) {
    return e->fail(TRG_ERR_INVALID_ARG, "map has non-finite coordinates");
  }
  // cell size = robot_size: a collision disc touches a 3x3 block, an edge ellipse <= 7x7
  float g = e->prm.robot_size;
  if (!(g > 0)) g = 0.3f;
  const double max_cells = 64.0 * 1024 * 1024;
  while (((double)(x1 - x0) / g + 2) * ((double)(y1 - y0) / g + 2) > max_cells) g *= 2.0f;
  const float inv_g = 1.0f / g;
  const int W = (int)floorf((x1 - x0) * inv_g) + 1;
  const int H = (int)floorf((y1 - y0) * inv_g) + 1;
  const size_t ncell = (size_t)W * H;

  if (m.cap_pts < n) {
    if (m.x) (void)hipFree(m.x);
    if (m.y) (void)hipFree(m.y);
    if (m.z) (void)hipFree(m.z);
    if (m.perm) (void)hipFree(m.perm);
    if (m.pt) (void)hipFree(m.pt);
    m.x = m.y = m.z = nullptr;
    m.perm = nullptr;
    m.pt = nullptr;
    HIPCHK(e, hipMalloc((void **)&m.pt, n * sizeof(float4)));
    HIPCHK(e, hipMalloc((void **)&m.x, n * sizeof(float)));
    HIPCHK(e, hipMalloc((void **)&m.y, n * sizeof(float)));
    HIPCHK(e, hipMalloc((void **)&m.z, n * sizeof(float)));
    HIPCHK(e, hipMalloc((void **)&m.perm, n * sizeof(int)));
    m.cap_pts = n;
  }
  if (m.cap_cells < ncell + 1) {
    if (m.cell_start) (void)hipFree(m.cell_start);
    m.cell_start = nullptr;
    HIPCHK(e, hipMalloc((void **)&m.cell_start, (ncell + 1) * sizeof(int)));
    m.cap_cells = ncell + 1;
  }
  // scratch of the build, kept with the engine (allocating and freeing 240 MB per build costs as much as
  // a kernel of it)
  IndexScratch &sc = e->idx_scratch;
  if (sc.cap_pts < n) {
    if (sc.aos) (void)hipFree(sc.aos);
    sc.aos = nullptr;
    sc.cap_pts = 0;
    HIPCHK(e, hipMalloc((void **)&sc.aos, n * 16));
    sc.cap_pts = n;
  }
  int bin_shift = 0, nbins = 0, nwg = 0;
  if (!getenv("TRG_INDEX_DIRECT") && index_bins_plan(n, ncell, &bin_shift, &nbins, &nwg)) {
    // through bins of ~one cell row (trg_kernels.hip): no global atomics, no random line per point
    const size_t nb = (size_t)nbins * nwg;
    if (sc.cap_bins < nb) {
      if (sc.hist) (void)hipFree(sc.hist);
      if (sc.base) (void)hipFree(sc.base);
      if (sc.bin_tmp) (void)hipFree(sc.bin_tmp);
      sc.hist = sc.base = sc.bin_tmp = nullptr;
      sc.cap_bins = 0;
      HIPCHK(e, hipMalloc((void **)&sc.hist, (nb + 1) * sizeof(int)));
      HIPCHK(e, hipMalloc((void **)&sc.base, (nb + 1) * sizeof(int)));
      HIPCHK(e, hipMalloc((void **)&sc.bin_tmp, (nb / 2048 + 4) * sizeof(int)));
      sc.cap_bins = nb;
    }
    // (the map's own record array is the first scratch: it is rewritten by the last kernel)
    launch_index_bins(d_xyz, n, stride, x0, y0, inv_g, W, H, (int)ncell, bin_shift, nbins, nwg, sc.hist, sc.base,
                      sc.bin_tmp, m.pt, (float4 *)sc.aos, m.cell_start, m.x, m.y, m.z, m.perm, m.pt, s);
  } else {
    if (sc.cap_direct < n) {
      if (sc.cell_of) (void)hipFree(sc.cell_of);
      if (sc.rank) (void)hipFree(sc.rank);
      sc.cell_of = sc.rank = nullptr;
      sc.cap_direct = 0;
      HIPCHK(e, hipMalloc((void **)&sc.cell_of, n * sizeof(int)));
      HIPCHK(e, hipMalloc((void **)&sc.rank, n * sizeof(int)));
      sc.cap_direct = n;
    }
    if (sc.cap_cells < ncell) {
      if (sc.counts) (void)hipFree(sc.counts);
      if (sc.tmp) (void)hipFree(sc.tmp);
      sc.counts = sc.tmp = nullptr;
      sc.cap_cells = 0;
      HIPCHK(e, hipMalloc((void **)&sc.counts, ncell * sizeof(int)));
      HIPCHK(e, hipMalloc((void **)&sc.tmp, (ncell / 2048 + 4) * sizeof(int)));
      sc.cap_cells = ncell;
    }
    int *d_cell_of = sc.cell_of, *d_rank = sc.rank, *d_counts = sc.counts, *d_tmp = sc.tmp;
    HIPCHK(e, hipMemsetAsync(d_counts, 0, ncell * sizeof(int), s));
    launch_cell_count(d_xyz, n, stride, x0, y0, inv_g, W, H, d_cell_of, d_rank, d_counts, s);
    launch_exclusive_scan(d_counts, m.cell_start, (int)ncell, d_tmp, s);
    launch_scatter_sort_aos(d_xyz, n, stride, d_cell_of, d_rank, (int)ncell, m.cell_start, sc.aos, m.x, m.y, m.z,
                            m.perm, m.pt, s);
  }
  HIPCHK(e, hipEventRecord(ev1, s));
  HIPCHK(e, hipStreamSynchronize(s));
  HIPCHK(e, hipGetLastError());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ev0, ev1);
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);

  m.n = n;
  m.g = g;
  m.bounds[0] = x0;
  m.bounds[1] = y0;
  m.bounds[2] = x1;
  m.bounds[3] = y1;
  m.view.x = m.x;
  m.view.y = m.y;
  m.view.z = m.z;
  m.view.pt = m.pt;
  m.view.perm = m.perm;
  m.view.cell_start = m.cell_start;
  m.view.x0 = x0;
  m.view.y0 = y0;
  m.view.inv_g = inv_g;
  m.view.W = W;
  m.view.H = H;
  m.view.n = (int)n;
  m.valid = true;
  m.top_wait();
  m.top_m = 0;  // (the top of the insertion tree: on demand, for the global map beside the build)
  if (&m == &e->gmap) {
    e->stats.map_points = n;
    e->stats.ms_index_build = ms;
    // read xyz once, write the cell-sorted SoA once, cell ids once
    e->stats.bytes_index_build = (uint64_t)(12 + 12 + 4) * n;
    e->stats.ms_set_map_total = ms_since(t_host);
    start_map_top(e, m);
  }
  return TRG_OK;
}

// Host cloud -> HBM.  TRG::setGlobalMap / setLocalMap get a cloud in ordinary (pageable) host memory
// (trg.cpp:179-193, 195-209); a plain hipMemcpy from there runs at a third of the link rate (the runtime
// stages it through one pinned buffer on one thread: ~25 ms for the 120 MB of C3).  Here UP_THREADS host
// threads copy interleaved chunks into pinned staging slots of their own and send every chunk on with
// hipMemcpyAsync on a stream of their own, so the CPU copies and the DMA transfers overlap; a source that
// is already pinned (hipHostMalloc / hipHostRegister / a pinned torch tensor) goes out in one async copy.
constexpr int UP_THREADS = 4, UP_SLOTS = 2;
constexpr size_t UP_CHUNK = (size_t)8 << 20;
struct Uploader {
  char *pinned = nullptr;  // UP_THREADS * UP_SLOTS chunks
  hipStream_t st[UP_THREADS] = {};
  hipEvent_t ev[UP_THREADS][UP_SLOTS] = {};
  float *d_in = nullptr;   // device staging of the raw cloud (kept across calls)
  size_t d_cap = 0;
  void release() {
    if (pinned) (void)hipHostFree(pinned);
    for (auto &s : st)
      if (s) (void)hipStreamDestroy(s);
    for (auto &row : ev)
      for (auto &x : row)
        if (x) (void)hipEventDestroy(x);
    if (d_in) (void)hipFree(d_in);
    *this = Uploader();
  }
};

TrgStatus staged_upload(TrgEngine *e, void *d_dst, const void *src, size_t bytes) {
  Uploader &u = *e->uploader;
  hipPointerAttribute_t attr;
  const bool pinned_src = hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost;
  (void)hipGetLastError();  // (an ordinary malloc pointer makes the query fail: not an error)
  if (pinned_src || bytes < UP_CHUNK) {
    HIPCHK(e, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, e->s_main));
    HIPCHK(e, hipStreamSynchronize(e->s_main));
    return TRG_OK;
  }
  if (!u.pinned) {
    HIPCHK(e, hipHostMalloc((void **)&u.pinned, UP_CHUNK * UP_THREADS * UP_SLOTS, hipHostMallocDefault));
    for (int t = 0; t < UP_THREADS; ++t) {
      HIPCHK(e, hipStreamCreateWithFlags(&u.st[t], hipStreamNonBlocking));
      for (int k = 0; k < UP_SLOTS; ++k) HIPCHK(e, hipEventCreateWithFlags(&u.ev[t][k], hipEventDisableTiming));
    }
  }
  const size_t nchunk = (bytes + UP_CHUNK - 1) / UP_CHUNK;
  std::atomic<int> bad{0};
  auto work = [&](int t) {
    if (hipSetDevice(e->device) != hipSuccess) {
      bad = 1;
      return;
    }
    int use = 0;
    for (size_t c = (size_t)t; c < nchunk; c += UP_THREADS, ++use) {
      const int k = use % UP_SLOTS;
      char *slot = u.pinned + ((size_t)t * UP_SLOTS + k) * UP_CHUNK;
      if (use >= UP_SLOTS && hipEventSynchronize(u.ev[t][k]) != hipSuccess) bad = 1;  // the slot's last transfer
      const size_t off = c * UP_CHUNK, len = std::min(UP_CHUNK, bytes - off);
      memcpy(slot, (const char *)src + off, len);
      if (hipMemcpyAsync((char *)d_dst + off, slot, len, hipMemcpyHostToDevice, u.st[t]) != hipSuccess) bad = 1;
      if (hipEventRecord(u.ev[t][k], u.st[t]) != hipSuccess) bad = 1;
    }
    if (hipStreamSynchronize(u.st[t]) != hipSuccess) bad = 1;
  };
  std::vector<std::thread> thr;
  for (int t = 1; t < UP_THREADS; ++t) thr.emplace_back(work, t);
  work(0);
  for (auto &th : thr) th.join();
  if (bad) return e->fail(TRG_ERR_DEVICE, "staged upload of the cloud failed");
  return TRG_OK;
}

TrgStatus upload_and_build(TrgEngine *e, DevMap &m, const float *xyz, size_t n, size_t stride) {
  if (n == 0) {
    m.n = 0;
    m.valid = false;
    return TRG_OK;
  }
  Uploader &u = *e->uploader;
  const size_t floats = n * stride;
  if (u.d_cap < floats) {
    if (u.d_in) (void)hipFree(u.d_in);
    u.d_in = nullptr;
    u.d_cap = 0;
    HIPCHK(e, hipMalloc((void **)&u.d_in, floats * sizeof(float)));
    u.d_cap = floats;
  }
  auto t0 = Clock::now();
  TrgStatus st = staged_upload(e, u.d_in, xyz, floats * sizeof(float));
  e->stats.ms_upload = ms_since(t0);
  if (st != TRG_OK) return st;
  return build_map(e, m, u.d_in, n, stride);
}

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

// ---- synchronous probes ------------------------------------------------------------------------
TrgStatus ensure_sync_scratch(TrgEngine *e, size_t m) {
  if (e->sy_cap >= m) return TRG_OK;
  size_t cap = std::max<size_t>(m, 1024);
  HIPCHK(e, alloc_pinned(e->sy_in, cap * 3));
  HIPCHK(e, alloc_pinned(e->sy_in2, cap * 3));
  HIPCHK(e, alloc_pinned(e->sy_f0, cap));
  HIPCHK(e, alloc_pinned(e->sy_f1, cap));
  HIPCHK(e, alloc_pinned(e->sy_i0, cap));
  HIPCHK(e, alloc_pinned(e->sy_i1, cap));
  HIPCHK(e, alloc_pinned(e->sy_i2, cap));
  if (e->sy_mid) (void)hipFree(e->sy_mid);
  e->sy_mid = nullptr;
  HIPCHK(e, hipMalloc((void **)&e->sy_mid, edge_mid_floats(cap) * sizeof(float)));
  e->sy_cap = cap;
  return TRG_OK;
}

DevMap *pick_map(TrgEngine *e, TrgKind k) { return k == TRG_KIND_LOCAL ? &e->lmap : &e->gmap; }

// strm: the stream the probe runs in (the main stream may still hold look-ahead work of a finished replay that
// nobody needs to wait for: the maps are read-only here)
// radius > 0: probe discs of that radius instead of robot_size (setLocalGraph / isFrontier ask for
// robot_size / 2, trg.cpp:214, 791) -- passed in the query parameters, the engine's own stay untouched
TrgStatus collision_sync(TrgEngine *e, DevMap &m, float threshold, const float *xy, size_t cnt,
                         int32_t *flag, int32_t *c_out, int32_t *n_out, hipStream_t strm = nullptr,
                         float radius = 0.0f) {
  if (!strm) strm = e->s_main;
  QueryParams qp = qparams(e);
  if (radius > 0.0f) qp.robot_size = radius;
  if (!m.valid) {
    // empty map: kd_nearest_range on an empty tree returns no hits -> collision (trg.cpp:749-752)
    for (size_t i = 0; i < cnt; ++i) {
      if (flag) flag[i] = 1;
      if (c_out) c_out[i] = 0;
      if (n_out) n_out[i] = 0;
    }
    return TRG_OK;
  }
  const size_t B = 1 << 20;
  for (size_t off = 0; off < cnt; off += B) {
    const size_t m_ = std::min(B, cnt - off);
    TrgStatus st = ensure_sync_scratch(e, m_);
    if (st != TRG_OK) return st;
    memcpy(e->sy_in.h, xy + 2 * off, m_ * 2 * sizeof(float));
    HIPCHK(e, hipMemcpyAsync(e->sy_in.d, e->sy_in.h, m_ * 2 * sizeof(float), hipMemcpyHostToDevice,
                             strm));
    launch_probe_collision(m.view, qp, threshold, e->sy_in.d, (int)m_, e->sy_i0.d,
                           e->sy_i1.d, e->sy_i2.d, e->d_ctr, strm);
    HIPCHK(e, hipMemcpyAsync(e->sy_i0.h, e->sy_i0.d, m_ * sizeof(int), hipMemcpyDeviceToHost,
                             strm));
    HIPCHK(e, hipMemcpyAsync(e->sy_i1.h, e->sy_i1.d, m_ * sizeof(int), hipMemcpyDeviceToHost,
                             strm));
    HIPCHK(e, hipMemcpyAsync(e->sy_i2.h, e->sy_i2.d, m_ * sizeof(int), hipMemcpyDeviceToHost,
                             strm));
    HIPCHK(e, hipStreamSynchronize(strm));
    HIPCHK(e, hipGetLastError());
    if (flag) memcpy(flag + off, e->sy_i0.h, m_ * sizeof(int));
    if (c_out) memcpy(c_out + off, e->sy_i1.h, m_ * sizeof(int));
    if (n_out) memcpy(n_out + off, e->sy_i2.h, m_ * sizeof(int));
  }
  e->stats.sync_batches++;
  return TRG_OK;
}

// ---- exact nearest-map-point tie-break ------------------------------------------------------------
// addNode takes the z of kd_nearest's result (trg.cpp:244-247).  When several map points are at the
// same minimal fp32 distance, kd_nearest returns the one it visits first (strict `<`, kdtree.c:343;
// the root is the initial best, kdtree.c:393-396), which depends on the shape of the insertion-built
// map tree.  The tree is never built here: as in kd_first_of_two (host_index.h) the visiting order
// of two tied points is decided at their lowest common ancestor, and the common path is followed
// by asking the GPU for "the point with the smallest original index inside this half-open region,
// inserted after the current ancestor" -- which is exactly the root of that subtree, because
// kd_insert appends in cloud order and sends `<` to the left (kdtree.c:179-198).
struct TiePoint {
  int perm;
  float x, y;
};

TrgStatus ensure_tie_scratch(TrgEngine *e) {
  if (e->mt_set_d) return TRG_OK;
  HIPCHK(e, hipMalloc((void **)&e->mt_set_d, sizeof(MapTieSet)));
  HIPCHK(e, hipMalloc((void **)&e->mt_walk_d, sizeof(MapTieWalk)));
  HIPCHK(e, hipHostMalloc((void **)&e->mt_set_h, sizeof(MapTieSet), hipHostMallocDefault));
  HIPCHK(e, hipHostMalloc((void **)&e->mt_walk_h, sizeof(MapTieWalk), hipHostMallocDefault));
  return TRG_OK;
}

// which of the tied points A, B the nearest-neighbour search for q visits first: 0 = A, 1 = B.  The
// walk (region scan + decision per tree level, ~30-50 levels on a 10 M-point map) runs on the device
// without the host in between: the first steps (regions of millions of points) one grid-wide kernel
// each, enqueued blindly, the rest inside a single workgroup; steps after the decision return at once.
constexpr int MAP_TOP_POINTS = 8192;  // points of the host-side top of the map tree

// The first MAP_TOP_POINTS points of the cloud, inserted like kd_insert does (kdtree.c:179-198: `<` goes
// left, the axis alternates with the depth): the top of the reference's map tree, node k = cloud point k.
static void insert_map_top(DevMap &m, int M) {
  m.top_left.assign(M, -1);
  m.top_right.assign(M, -1);
  for (int k = 1; k < M; ++k) {
    const float px = m.top_xy[2 * (size_t)k], py = m.top_xy[2 * (size_t)k + 1];
    int cur = 0, axis = 0;
    for (;;) {
      const float split = axis ? m.top_xy[2 * (size_t)cur + 1] : m.top_xy[2 * (size_t)cur];
      int &child = ((axis ? py : px) < split) ? m.top_left[cur] : m.top_right[cur];
      if (child < 0) {
        child = k;
        break;
      }
      cur = child;
      axis ^= 1;
    }
  }
  m.top_m = M;
}

TrgStatus ensure_map_top(TrgEngine *e, DevMap &m) {
  m.top_wait();  // (the global map's top is prepared beside the build)
  if (m.top_m > 0) return TRG_OK;
  const int M = (int)std::min<size_t>(m.n, MAP_TOP_POINTS);
  float *d_xy = nullptr;
  HIPCHK(e, hipMalloc((void **)&d_xy, (size_t)M * 2 * sizeof(float)));
  launch_collect_first(m.view, M, d_xy, e->s_aux);
  m.top_xy.resize((size_t)M * 2);
  hipError_t he = hipMemcpyAsync(m.top_xy.data(), d_xy, (size_t)M * 2 * sizeof(float), hipMemcpyDeviceToHost, e->s_aux);
  if (he == hipSuccess) he = hipStreamSynchronize(e->s_aux);
  (void)hipFree(d_xy);
  if (he != hipSuccess) return e->fail(TRG_ERR_DEVICE, std::string("map top: ") + hipGetErrorString(he));
  insert_map_top(m, M);
  return TRG_OK;
}

// The same beside the build: the tie-breaking scratch is allocated, the first points are fetched on the aux
// stream, and a helper thread waits for them and inserts them while the graph is being built -- the first
// nearest-point tie of a build otherwise paid ~0.9 ms for all of this inside the level loop.  Failures are
// silent here: ensure_map_top then does the work on demand.
void start_map_top(TrgEngine *e, DevMap &m) {
  m.top_wait();
  m.top_m = 0;
  if (m.n == 0) return;
  if (ensure_tie_scratch(e) != TRG_OK) return;
  const int M = (int)std::min<size_t>(m.n, MAP_TOP_POINTS);
  if (!e->top_xy_d && hipMalloc((void **)&e->top_xy_d, (size_t)MAP_TOP_POINTS * 2 * sizeof(float)) != hipSuccess) return;
  if (!e->top_xy_h &&
      hipHostMalloc((void **)&e->top_xy_h, (size_t)MAP_TOP_POINTS * 2 * sizeof(float), hipHostMallocDefault) != hipSuccess)
    return;
  if (!e->top_ev && hipEventCreateWithFlags(&e->top_ev, hipEventDisableTiming) != hipSuccess) return;
  launch_collect_first(m.view, M, e->top_xy_d, e->s_aux);
  if (hipMemcpyAsync(e->top_xy_h, e->top_xy_d, (size_t)M * 2 * sizeof(float), hipMemcpyDeviceToHost, e->s_aux) !=
          hipSuccess ||
      hipEventRecord(e->top_ev, e->s_aux) != hipSuccess)
    return;
  const int dev = e->device;
  hipEvent_t ev = e->top_ev;
  const float *src = e->top_xy_h;
  DevMap *mp = &m;
  m.top_thread = std::thread([dev, ev, src, mp, M] {
    (void)hipSetDevice(dev);
    if (hipEventSynchronize(ev) != hipSuccess) return;
    mp->top_xy.assign(src, src + (size_t)M * 2);
    insert_map_top(*mp, M);
  });
}

TrgStatus map_first_of_two(TrgEngine *e, DevMap &m, float qx, float qy, const TiePoint &A,
                           const TiePoint &B, int *first) {
  hipStream_t s = e->s_aux;  // (the main stream is busy with the next level's speculative expansion)
  TrgStatus st = ensure_map_top(e, m);
  if (st != TRG_OK) return st;
  MapTieWalk w{};
  w.key = ~0ull;
  w.lo[0] = w.lo[1] = -INFINITY;
  w.hi[0] = w.hi[1] = INFINITY;
  w.cur_perm = -1;
  w.axis = 0;
  w.qx = qx;
  w.qy = qy;
  w.aperm = A.perm;
  w.bperm = B.perm;
  w.ax = A.x;
  w.ay = A.y;
  w.bx = B.x;
  w.by = B.y;
  // the common path of A and B through the top of the tree, on the host (the same decisions as
  // region_step on the device, trg_kernels.hip)
  {
    int cur = 0;
    for (;;) {
      const float cx = m.top_xy[2 * (size_t)cur], cy = m.top_xy[2 * (size_t)cur + 1];
      const int axis = w.axis;
      const float split = axis ? cy : cx;
      const float q = axis ? qy : qx;
      const bool near_is_left = (q - split) <= 0;
      const float ca = axis ? A.y : A.x, cb = axis ? B.y : B.x;
      if (cur == A.perm || cur == B.perm) {
        const bool cur_is_a = cur == A.perm;
        const bool other_left = (cur_is_a ? cb : ca) < split;
        const bool other_first = other_left == near_is_left;
        *first = cur_is_a ? (other_first ? 1 : 0) : (other_first ? 0 : 1);
        return TRG_OK;
      }
      const bool a_left = ca < split, b_left = cb < split;
      if (a_left != b_left) {
        *first = (a_left == near_is_left) ? 0 : 1;
        return TRG_OK;
      }
      if (a_left)
        w.hi[axis] = split;
      else
        w.lo[axis] = split;
      w.cur_perm = cur;
      w.axis = axis ^ 1;
      w.steps++;
      const int child = a_left ? m.top_left[cur] : m.top_right[cur];
      if (child < 0) break;  // the subtree's root is a later point: the device goes on from this region
      cur = child;
    }
  }
  *e->mt_walk_h = w;
  HIPCHK(e, hipMemcpyAsync(e->mt_walk_d, e->mt_walk_h, sizeof(MapTieWalk), hipMemcpyHostToDevice, s));
  for (int batch = 0; batch < 64; ++batch) {
    // what is left of the region after the top of the tree holds ~N / 8192 points: one workgroup walks it
    // (a full-size region -- a map smaller than the top -- cannot get here)
    launch_map_tie_walk(m.view, e->mt_walk_d, 0, 64, s);
    HIPCHK(e, hipMemcpyAsync(e->mt_walk_h, e->mt_walk_d, sizeof(MapTieWalk), hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipStreamSynchronize(s));
    if (e->mt_walk_h->done == 1) {
      *first = e->mt_walk_h->first;
      return TRG_OK;
    }
    if (e->mt_walk_h->done) break;
  }
  return e->fail(TRG_ERR_DEVICE, "nearest-point tie-break lost its candidates (internal error)");
}

// z of the map point kd_nearest returns for (qx, qy), ties decided as the reference's tree does
TrgStatus map_nn_exact(TrgEngine *e, DevMap &m, float qx, float qy, float *z, bool *found) {
  TrgStatus st = ensure_tie_scratch(e);
  if (st != TRG_OK) return st;
  hipStream_t s = e->s_aux;  // (the map is read-only here; the main stream may hold speculative work)
  launch_map_tied_set(m.view, qx, qy, e->prm.robot_size, e->mt_set_d, s);
  HIPCHK(e, hipMemcpyAsync(e->mt_set_h, e->mt_set_d, sizeof(MapTieSet), hipMemcpyDeviceToHost, s));
  HIPCHK(e, hipStreamSynchronize(s));
  const MapTieSet T = *e->mt_set_h;
  *found = T.count > 0;
  if (!*found) return TRG_OK;
  const int n = std::min(T.count, MAPTIE_SET_CAP);
  // lowest original index first, so that an unresolved case equals the hot kernels' provisional pick
  int order[MAPTIE_SET_CAP];
  for (int i = 0; i < n; ++i) order[i] = i;
  std::sort(order, order + n, [&](int a, int b) { return T.perm[a] < T.perm[b]; });
  *z = T.z[order[0]];
  if (T.count == 1) return TRG_OK;
  if (T.count > MAPTIE_SET_CAP) {
    e->stats.map_nn_unresolved++;
    return TRG_OK;
  }
  e->stats.map_nn_resolved++;
  bool same_z = true;
  for (int i = 1; i < n; ++i) same_z = same_z && T.z[order[i]] == T.z[order[0]];
  if (same_z) return TRG_OK;
  if (T.perm[order[0]] == 0) return TRG_OK;  // the root keeps an equal distance (kdtree.c:393-396)
  int w = order[0];
  for (int i = 1; i < n; ++i) {
    const int c = order[i];
    const TiePoint A{T.perm[w], T.x[w], T.y[w]}, B{T.perm[c], T.x[c], T.y[c]};
    int first = 0;
    st = map_first_of_two(e, m, qx, qy, A, B, &first);
    if (st != TRG_OK) return st;
    if (first == 1) w = c;
  }
  *z = T.z[w];
  return TRG_OK;
}

TrgStatus nearest_z_sync(TrgEngine *e, DevMap &m, const float *xy, size_t cnt, float *z,
                         int32_t *found) {
  if (!m.valid) return e->fail(TRG_ERR_NO_MAP, "nearest_z on an empty map");
  const size_t B = 1 << 20;
  for (size_t off = 0; off < cnt; off += B) {
    const size_t m_ = std::min(B, cnt - off);
    TrgStatus st = ensure_sync_scratch(e, m_);
    if (st != TRG_OK) return st;
    memcpy(e->sy_in.h, xy + 2 * off, m_ * 2 * sizeof(float));
    HIPCHK(e, hipMemcpyAsync(e->sy_in.d, e->sy_in.h, m_ * 2 * sizeof(float), hipMemcpyHostToDevice,
                             e->s_main));
    launch_probe_nearest_z(m.view, qparams(e), e->sy_in.d, (int)m_, e->sy_f0.d, e->sy_i0.d,
                           e->d_ctr, e->s_main);
    HIPCHK(e, hipMemcpyAsync(e->sy_f0.h, e->sy_f0.d, m_ * sizeof(float), hipMemcpyDeviceToHost,
                             e->s_main));
    HIPCHK(e, hipMemcpyAsync(e->sy_i0.h, e->sy_i0.d, m_ * sizeof(int), hipMemcpyDeviceToHost,
                             e->s_main));
    HIPCHK(e, hipStreamSynchronize(e->s_main));
    HIPCHK(e, hipGetLastError());
    memcpy(z + off, e->sy_f0.h, m_ * sizeof(float));
    // found == 2: several map points at the same fp32 distance; ask for the reference's choice
    std::vector<size_t> tied;
    for (size_t i = 0; i < m_; ++i) {
      if (e->sy_i0.h[i] == 2) tied.push_back(i);
      if (found) found[off + i] = e->sy_i0.h[i] ? 1 : 0;
    }
    for (size_t i : tied) {
      bool f = false;
      float zz = 0;
      st = map_nn_exact(e, m, xy[2 * (off + i)], xy[2 * (off + i) + 1], &zz, &f);
      if (st != TRG_OK) return st;
      if (f) z[off + i] = zz;
    }
  }
  e->stats.sync_batches++;
  return TRG_OK;
}

// The reference's slope gate (trg.cpp:269-274) evaluated with the host libm, used only for the
// sliver the device's exact rational test could not decide.
inline bool host_slope_gate(const TrgEngine *e, float z1, float z2, float dist) {
  float max_slope = atan2(e->prm.height_threshold, e->prm.robot_size);
  float slope = atan2(fabs(z1 - z2), dist);
  return slope > max_slope;
}
// final status code (0..4) of an edge evaluation after resolving an uncertain gate
inline int resolve_status(TrgEngine *e, int raw, float z1, float z2, float dist) {
  if (raw & EDGE_GATE_UNCERTAIN) {
    e->stats.gate_uncertain++;
    if (host_slope_gate(e, z1, z2, dist)) return EDGE_GATE;
  }
  return raw & EDGE_STATUS_MASK;
}

TrgStatus edges_sync(TrgEngine *e, DevMap &m, const float *p1, const float *p2, size_t cnt,
                     int32_t *status, int32_t *n_pts, float *weight, float *dist, bool resolve) {
  if (!m.valid) return e->fail(TRG_ERR_NO_MAP, "edge evaluation on an empty map");
  const size_t B = 1 << 18;
  for (size_t off = 0; off < cnt; off += B) {
    const size_t m_ = std::min(B, cnt - off);
    TrgStatus st = ensure_sync_scratch(e, m_);
    if (st != TRG_OK) return st;
    memcpy(e->sy_in.h, p1 + 3 * off, m_ * 3 * sizeof(float));
    memcpy(e->sy_in2.h, p2 + 3 * off, m_ * 3 * sizeof(float));
    HIPCHK(e, hipMemcpyAsync(e->sy_in.d, e->sy_in.h, m_ * 3 * sizeof(float), hipMemcpyHostToDevice,
                             e->s_main));
    HIPCHK(e, hipMemcpyAsync(e->sy_in2.d, e->sy_in2.h, m_ * 3 * sizeof(float),
                             hipMemcpyHostToDevice, e->s_main));
    launch_edges(m.view, qparams(e), e->sy_in.d, e->sy_in2.d, (int)m_, e->sy_mid, e->sy_i0.d,
                 e->sy_i1.d, e->sy_f0.d, e->sy_f1.d, e->d_ctr, e->s_main);
    HIPCHK(e, hipMemcpyAsync(e->sy_i0.h, e->sy_i0.d, m_ * sizeof(int), hipMemcpyDeviceToHost,
                             e->s_main));
    HIPCHK(e, hipMemcpyAsync(e->sy_i1.h, e->sy_i1.d, m_ * sizeof(int), hipMemcpyDeviceToHost,
                             e->s_main));
    HIPCHK(e, hipMemcpyAsync(e->sy_f0.h, e->sy_f0.d, m_ * sizeof(float), hipMemcpyDeviceToHost,
                             e->s_main));
    HIPCHK(e, hipMemcpyAsync(e->sy_f1.h, e->sy_f1.d, m_ * sizeof(float), hipMemcpyDeviceToHost,
                             e->s_main));
    HIPCHK(e, hipStreamSynchronize(e->s_main));
    HIPCHK(e, hipGetLastError());
    for (size_t i = 0; i < m_; ++i) {
      int raw = e->sy_i0.h[i];
      int stt = raw;
      if (resolve) {
        stt = resolve_status(e, raw, p1[3 * (off + i) + 2], p2[3 * (off + i) + 2], e->sy_f1.h[i]);
      }
      if (status) status[off + i] = stt;
      if (n_pts) n_pts[off + i] = e->sy_i1.h[i];
      if (weight) weight[off + i] = (stt == EDGE_OK) ? e->sy_f0.h[i] : 0.0f;
      if (dist) dist[off + i] = e->sy_f1.h[i];
    }
    e->stats.edge_evals_gpu += m_;
  }
  e->stats.sync_batches++;
  return TRG_OK;
}

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

// ---- chunk pipeline ----------------------------------------------------------------------------
TrgStatus ensure_chunks(TrgEngine *e) {
  const int S = e->prm.sample_num;
  if (e->chunk_S == S && e->chunks[0].done) return TRG_OK;
  const size_t cmax = TrgEngine::CHUNK_MAX;
  const size_t slots = cmax * (size_t)std::max(S, 1);
  Chunk *all_chunks[TrgEngine::NCHUNK + 1];
  for (int i = 0; i < TrgEngine::NCHUNK; ++i) all_chunks[i] = &e->chunks[i];
  all_chunks[TrgEngine::NCHUNK] = &e->root_chunk;
  for (Chunk *cp : all_chunks) {
    Chunk &c = *cp;
    if (!c.done) {
      HIPCHK(e, hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
      HIPCHK(e, hipEventCreate(&c.t0));
      HIPCHK(e, hipEventCreate(&c.t1));
      HIPCHK(e, hipEventCreate(&c.t2));
    }
    HIPCHK(e, alloc_pinned(c.in_blob, 6 * cmax));
    HIPCHK(e, alloc_pinned(c.out_blob, 2 * cmax + 6 * slots));
    HIPCHK(e, alloc_pinned(c.mt, 4 + 4 * (size_t)MAPTIE_CAP));
    if (c.mid_cap < slots) {
      if (c.d_mid) (void)hipFree(c.d_mid);
      c.d_mid = nullptr;
      HIPCHK(e, hipMalloc((void **)&c.d_mid, edge_mid_floats(slots) * sizeof(float)));
      c.mid_cap = slots;
    }
  }
  for (EdgeBatch &b : e->ebatches) {
    if (!b.done) {
      HIPCHK(e, hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
      HIPCHK(e, hipEventCreate(&b.t0));
      HIPCHK(e, hipEventCreate(&b.t1));
    }
    HIPCHK(e, alloc_pinned(b.p1, (size_t)TrgEngine::EBATCH_MAX * 3));
    HIPCHK(e, alloc_pinned(b.p2, (size_t)TrgEngine::EBATCH_MAX * 3));
    HIPCHK(e, alloc_pinned(b.weight, (size_t)TrgEngine::EBATCH_MAX));
    HIPCHK(e, alloc_pinned(b.dist, (size_t)TrgEngine::EBATCH_MAX));
    HIPCHK(e, alloc_pinned(b.status, (size_t)TrgEngine::EBATCH_MAX));
    if (!b.d_mid)
      HIPCHK(e, hipMalloc((void **)&b.d_mid,
                          edge_mid_floats(TrgEngine::EBATCH_MAX) * sizeof(float)));
  }
  e->chunk_S = S;
  return TRG_OK;
}

TrgStatus submit_chunk(TrgEngine *e, Chunk &c, int first, int count) {
  const int S = e->prm.sample_num;
  c.first = first;
  c.count = count;
  c.carve(count, S);
  for (int i = 0; i < count; ++i) {
    const int id = e->queue[first + i];
    c.node_xy.h[2 * i] = e->nx[id];
    c.node_xy.h[2 * i + 1] = e->ny[id];
    c.node_xyz.h[3 * i] = e->nx[id];
    c.node_xyz.h[3 * i + 1] = e->ny[id];
    c.node_xyz.h[3 * i + 2] = e->nz[id];
    c.node_id.h[i] = id;
  }
  hipStream_t s = e->s_main;
  HIPCHK(e, hipMemcpyAsync(c.in_blob.d, c.in_blob.h, c.in_words * sizeof(uint32_t),
                           hipMemcpyHostToDevice, s));
  const QueryParams q = qparams(e);
  HIPCHK(e, hipMemsetAsync(c.mt.d, 0, sizeof(int), s));
  HIPCHK(e, hipEventRecord(c.t0, s));
  launch_sample_nodes(e->gmap.view, q, e->d_cos, e->d_sin, e->sampler.table_bits, e->sampler.seed,
                      e->epoch, c.node_xy.d, c.node_id.d, count, c.n_acc.d, c.n_draws.d, c.sx.d,
                      c.sy.d, c.sz.d, e->d_ctr, c.mt.d, (MapTieRec *)(c.mt.d + 4), s);
  HIPCHK(e, hipEventRecord(c.t1, s));
  launch_spec_edges(e->gmap.view, q, c.node_xyz.d, count, c.n_acc.d, c.sx.d, c.sy.d, c.sz.d,
                    c.d_mid, c.status.d, nullptr, c.weight.d, c.dist.d, e->d_ctr, s);
  HIPCHK(e, hipEventRecord(c.t2, s));
  HIPCHK(e, hipMemcpyAsync(c.out_blob.h, c.out_blob.d, c.out_words * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, s));
  HIPCHK(e, hipMemcpyAsync(c.mt.h, c.mt.d, (4 + 4 * (size_t)MAPTIE_CAP) * sizeof(int),
                           hipMemcpyDeviceToHost, s));
  HIPCHK(e, hipEventRecord(c.done, s));
  c.in_flight = true;
  e->stats.launches_sample_kernel++;
  e->stats.launches_spec_kernel++;
  return TRG_OK;
}

// the same launch sequence for an explicit list of node ids (the roots of updateGraph's expansions)
TrgStatus submit_nodes(TrgEngine *e, Chunk &c, const int *ids, int count) {
  std::vector<int> saved;
  saved.swap(e->queue);
  e->queue.assign(ids, ids + count);
  const TrgStatus st = submit_chunk(e, c, 0, count);
  e->queue.swap(saved);
  return st;
}

TrgStatus wait_chunk(TrgEngine *e, Chunk &c) {
  if (!c.in_flight) return TRG_OK;
  auto t0 = Clock::now();
  HIPCHK(e, hipEventSynchronize(c.done));
  e->stats.ms_wait_gpu += ms_since(t0);
  float ms = 0;
  if (hipEventElapsedTime(&ms, c.t0, c.t1) == hipSuccess) e->stats.ms_sample_kernel += ms;
  if (hipEventElapsedTime(&ms, c.t1, c.t2) == hipSuccess) e->stats.ms_spec_kernel += ms;
  c.in_flight = false;
  // accepted samples whose elevation hung on a nearest-point tie: take the point the reference's
  // map tree returns, and re-evaluate the parent edge if that changed the sample's z
  const int n_mt = c.mt.h[0];
  if (n_mt > 0) {
    const int S = e->prm.sample_num;
    const MapTieRec *recs = (const MapTieRec *)(c.mt.h + 4);
    if (n_mt > MAPTIE_CAP) e->stats.map_nn_unresolved += (uint64_t)(n_mt - MAPTIE_CAP);
    for (int k = 0; k < std::min(n_mt, MAPTIE_CAP); ++k) {
      const MapTieRec &r = recs[k];
      float z = 0;
      bool found = false;
      TrgStatus st = map_nn_exact(e, e->gmap, r.qx, r.qy, &z, &found);
      if (st != TRG_OK) return st;
      if (!found || !(z != c.sz.h[r.slot])) continue;
      c.sz.h[r.slot] = z;
      const int qi = r.slot / S;
      const float p1[3] = {c.node_xyz.h[3 * qi], c.node_xyz.h[3 * qi + 1], c.node_xyz.h[3 * qi + 2]};
      const float p2[3] = {r.qx, r.qy, z};
      int32_t stt = 0;
      float w = 0, d = 0;
      st = edges_sync(e, e->gmap, p1, p2, 1, &stt, nullptr, &w, &d, true);
      if (st != TRG_OK) return st;
      c.status.h[r.slot] = stt;
      c.weight.h[r.slot] = w;
      c.dist.h[r.slot] = d;
    }
  }
  return TRG_OK;
}

TrgStatus collect_batch(TrgEngine *e, EdgeBatch &b) {
  if (!b.in_flight) return TRG_OK;
  auto t0 = Clock::now();
  HIPCHK(e, hipEventSynchronize(b.done));
  e->stats.ms_wait_gpu += ms_since(t0);
  float ms = 0;
  if (hipEventElapsedTime(&ms, b.t0, b.t1) == hipSuccess) e->stats.ms_edge_kernel += ms;
  for (int i = 0; i < b.count; ++i) {
    CallRec &c = e->calls[b.call_idx[i]];
    c.dist = b.dist.h[i];
    c.status = resolve_status(e, b.status.h[i], e->nz[c.n1], e->nz[c.n2], c.dist);
    c.weight = (c.status == EDGE_OK) ? b.weight.h[i] : 0.0f;
  }
  b.in_flight = false;
  b.count = 0;
  return TRG_OK;
}

// ship the pending deferred wireEdge evaluations (node -> already existing node) to the GPU
TrgStatus flush_pending(TrgEngine *e, bool all) {
  size_t pos = 0;
  while (e->pending_calls.size() - pos >= (all ? 1u : (size_t)TrgEngine::EBATCH_MAX)) {
    // find a free batch buffer, collecting the oldest if none
    EdgeBatch *b = nullptr;
    for (EdgeBatch &cand : e->ebatches)
      if (!cand.in_flight) {
        b = &cand;
        break;
      }
    if (!b) {
      TrgStatus st = collect_batch(e, e->ebatches[0]);
      if (st != TRG_OK) return st;
      // rotate so that [0] is again the oldest
      std::rotate(e->ebatches, e->ebatches + 1, e->ebatches + TrgEngine::NEBATCH);
      b = &e->ebatches[TrgEngine::NEBATCH - 1];
    }
    const int cnt = (int)std::min<size_t>(TrgEngine::EBATCH_MAX, e->pending_calls.size() - pos);
    b->call_idx.assign(e->pending_calls.begin() + pos, e->pending_calls.begin() + pos + cnt);
    for (int i = 0; i < cnt; ++i) {
      const CallRec &c = e->calls[b->call_idx[i]];
      b->p1.h[3 * i] = e->nx[c.n1];
      b->p1.h[3 * i + 1] = e->ny[c.n1];
      b->p1.h[3 * i + 2] = e->nz[c.n1];
      b->p2.h[3 * i] = e->nx[c.n2];
      b->p2.h[3 * i + 1] = e->ny[c.n2];
      b->p2.h[3 * i + 2] = e->nz[c.n2];
    }
    hipStream_t s = e->s_edge;
    HIPCHK(e, hipMemcpyAsync(b->p1.d, b->p1.h, (size_t)cnt * 3 * sizeof(float),
                             hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(b->p2.d, b->p2.h, (size_t)cnt * 3 * sizeof(float),
                             hipMemcpyHostToDevice, s));
    HIPCHK(e, hipEventRecord(b->t0, s));
    launch_edges(e->gmap.view, qparams(e), b->p1.d, b->p2.d, cnt, b->d_mid, b->status.d, nullptr,
                 b->weight.d, b->dist.d, e->d_ctr, s);
    HIPCHK(e, hipEventRecord(b->t1, s));
    HIPCHK(e, hipMemcpyAsync(b->status.h, b->status.d, (size_t)cnt * sizeof(int),
                             hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipMemcpyAsync(b->weight.h, b->weight.d, (size_t)cnt * sizeof(float),
                             hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipMemcpyAsync(b->dist.h, b->dist.d, (size_t)cnt * sizeof(float),
                             hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipEventRecord(b->done, s));
    b->in_flight = true;
    b->count = cnt;
    e->stats.launches_edge_kernel++;
    e->stats.edge_evals_gpu += cnt;
    pos += cnt;
  }
  e->pending_calls.erase(e->pending_calls.begin(), e->pending_calls.begin() + pos);
  if (all) {
    for (EdgeBatch &b : e->ebatches) {
      TrgStatus st = collect_batch(e, b);
      if (st != TRG_OK) return st;
    }
  }
  return TRG_OK;
}

inline void emit_deferred(TrgEngine *e, int n1, int n2) {
  if (n1 == n2) return;  // wireEdge returns at once (trg.cpp:255-257)
  e->calls.push_back(CallRec{n1, n2, -1, 0.0f, 0.0f});
  e->pending_calls.push_back((int)e->calls.size() - 1);
  e->stats.edge_calls++;
}

// Apply the logged wireEdge() calls in program order: the dedupe of trg.cpp:255-267 and the two
// push_backs of trg.cpp:365-368.  `from` = first call not yet applied.
void apply_calls(TrgEngine *e, size_t from) {
  e->edges.grow_nodes(e->nx.size());
  for (size_t i = from; i < e->calls.size(); ++i) {
    const CallRec &c = e->calls[i];
    if (c.n1 == c.n2) continue;
    if (e->edges.has(c.n1, c.n2) || e->edges.has(c.n2, c.n1)) continue;
    if (c.status != EDGE_OK) continue;
    e->edges.push(c.n1, c.n2, c.weight, c.dist);
    e->edges.push(c.n2, c.n1, c.weight, c.dist);
  }
}

// BFS expansion from the node `ref_id`, replaying trg.cpp:372-454 with GPU results.
// `applied` is the index of the first call not yet folded into e->edges; step 3's validity test
// needs edges of brand-new nodes only, which it derives locally.
// pre / pre_qi: GPU results of the root already fetched (entry pre_qi of chunk *pre, used by
// updateGraph, which fetches all its roots in bulk); the root then needs no round trip of its own.
TrgStatus expand_bfs(TrgEngine *e, int ref_id, Chunk *pre = nullptr, int pre_qi = 0) {
  TrgStatus st = ensure_chunks(e);
  if (st != TRG_OK) return st;
  const int S = e->prm.sample_num;
  const float r = e->prm.robot_size;
  e->queue.clear();
  e->queue.push_back(ref_id);
  size_t head = 0;        // next queue position to replay
  size_t submitted = pre ? 1 : 0;  // queue positions [0, submitted) have been shipped to the GPU
  int next_buf = 0;       // chunk buffers are used round-robin, so completion order == queue order
  std::deque<int> inflight;  // chunk buffer indices in submission order
  std::vector<int> range_hits;
  std::vector<float> s3_p1, s3_p2, s3_w, s3_d;
  std::vector<int32_t> s3_st;
  auto t_replay = Clock::now();
  double waited0 = e->stats.ms_wait_gpu;

  // Ship queue positions [submitted, submitted+cnt) in the next free buffer.  Buffers are used
  // round-robin and consumed in the same order, so the oldest in-flight chunk is always next.
  auto ship = [&](size_t cnt) -> TrgStatus {
    Chunk &c = e->chunks[next_buf];
    TrgStatus s2 = submit_chunk(e, c, (int)submitted, (int)cnt);
    if (s2 != TRG_OK) return s2;
    inflight.push_back(next_buf);
    next_buf = (next_buf + 1) % TrgEngine::NCHUNK;
    submitted += cnt;
    return TRG_OK;
  };

  Chunk *cur = nullptr;
  while (head < e->queue.size()) {
    // keep the GPU fed while the replay works: full-size chunks as soon as enough nodes are queued,
    // a small one only when the chunk being replayed is about to run dry
    for (;;) {
      const size_t avail = e->queue.size() - submitted;
      const int busy = (int)inflight.size() + (cur ? 1 : 0);
      if (avail == 0 || busy >= TrgEngine::NCHUNK) break;
      const size_t left = cur ? (size_t)(cur->first + cur->count) - head : 0;
      const bool starving = inflight.empty() && left <= 16;
      if (avail < 512 && !starving) break;
      st = ship(std::min<size_t>(avail, TrgEngine::CHUNK_MAX));
      if (st != TRG_OK) return st;
    }
    const bool use_pre = pre && head == 0;
    if (!use_pre && (!cur || (int)head >= cur->first + cur->count)) {
      cur = nullptr;
      if (inflight.empty()) return e->fail(TRG_ERR_DEVICE, "replay starved (internal error)");
      cur = &e->chunks[inflight.front()];
      inflight.pop_front();
      st = wait_chunk(e, *cur);
      if (st != TRG_OK) return st;
    }
    Chunk *const src = use_pre ? pre : cur;
    const int qi = use_pre ? pre_qi : (int)head - cur->first;
    const int node = e->queue[head];
    head++;
    e->stats.expanded_nodes++;
    const int n_acc = src->n_acc.h[qi];
    e->stats.trials += src->n_draws.h[qi];
    e->stats.samples += n_acc;
    e->stats.edge_evals_gpu += n_acc;

    for (int j = 0; j < n_acc; ++j) {
      const int slot = qi * S + j;
      const float sx = src->sx.h[slot], sy = src->sy.h[slot];
      // 1. nearest existing node (trg.cpp:408-417)
      const int ex = nearest_node(e, sx, sy);
      if (e->nstate[ex] == TRG_NODE_INVALID) continue;
      if (norm2f(e->nx[ex] - sx, e->ny[ex] - sy) < r) {
        emit_deferred(e, node, ex);
        continue;
      }
      // 2. new node (trg.cpp:420-426); its parent edge was evaluated speculatively on the GPU
      const int new_state = (ref_id == 0) ? TRG_NODE_VALID : TRG_NODE_FRONTIER;
      const float sz = src->sz.h[slot];
      const int nn = add_node_host(e, sx, sy, sz, new_state);
      const float dist = src->dist.h[slot];
      const int stt = resolve_status(e, src->status.h[slot], e->nz[node], sz, dist);
      const bool parent_ok = (stt == EDGE_OK);
      e->calls.push_back(
          CallRec{node, nn, stt, parent_ok ? src->weight.h[slot] : 0.0f, dist});
      e->stats.edge_calls++;
      bool has_edge = parent_ok;

      // 3. neighbour wiring (trg.cpp:429-444), only for configs like indoor.yaml
      if (e->step3) {
        kd_sync(e);
        e->kd.range(e->nx[nn], e->ny[nn], e->prm.expand_dist, range_hits);
        const size_t first_call = e->calls.size();
        for (int other : range_hits) {
          if (e->nstate[other] == TRG_NODE_INVALID) continue;
          emit_deferred(e, nn, other);
        }
        if (!parent_ok && e->calls.size() > first_call) {
          // the node's fate hangs on these edges: evaluate them now (synchronous round trip)
          const size_t m = e->calls.size() - first_call;
          s3_p1.resize(3 * m);
          s3_p2.resize(3 * m);
          s3_st.resize(m);
          s3_w.resize(m);
          s3_d.resize(m);
          for (size_t k = 0; k < m; ++k) {
            const CallRec &c = e->calls[first_call + k];
            s3_p1[3 * k] = e->nx[c.n1];
            s3_p1[3 * k + 1] = e->ny[c.n1];
            s3_p1[3 * k + 2] = e->nz[c.n1];
            s3_p2[3 * k] = e->nx[c.n2];
            s3_p2[3 * k + 1] = e->ny[c.n2];
            s3_p2[3 * k + 2] = e->nz[c.n2];
          }
          st = edges_sync(e, e->gmap, s3_p1.data(), s3_p2.data(), m, s3_st.data(), nullptr,
                          s3_w.data(), s3_d.data(), true);
          if (st != TRG_OK) return st;
          for (size_t k = 0; k < m; ++k) {
            CallRec &c = e->calls[first_call + k];
            c.status = s3_st[k];
            c.weight = s3_w[k];
            c.dist = s3_d[k];
            if (c.status == EDGE_OK) has_edge = true;
          }
          // they are resolved: take them off the pending list (they were appended last)
          e->pending_calls.resize(e->pending_calls.size() - m);
        }
      }

      // 4. (trg.cpp:447-451)
      if (!has_edge) {
        e->nstate[nn] = TRG_NODE_INVALID;
        e->stats.invalid_nodes++;
        continue;
      }
      e->queue.push_back(nn);
    }
    if ((int)e->pending_calls.size() >= TrgEngine::EBATCH_MAX) {
      st = flush_pending(e, false);
      if (st != TRG_OK) return st;
    }
  }
  e->stats.ms_replay_host += ms_since(t_replay) - (e->stats.ms_wait_gpu - waited0);
  return TRG_OK;
}

// ---- cleanGraph (trg.cpp:491-535) ---------------------------------------------------------------
void snapshot_csr(const TrgEngine *e, Csr &out) {
  const size_t V = e->nx.size();
  out.clear();
  out.xyz.resize(3 * V);
  out.state.resize(V);
  out.cid.resize(V);
  out.rowptr.resize(V + 1);
  out.rowptr[0] = 0;
  for (size_t i = 0; i < V; ++i) out.rowptr[i + 1] = out.rowptr[i] + (i < e->edges.deg.size() ? e->edges.deg[i] : 0);
  const size_t E = out.rowptr[V];
  out.col.resize(E);
  out.w.resize(E);
  out.dist.resize(E);
  parallel_ranges(V, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; ++i) {
      out.xyz[3 * i] = e->nx[i];
      out.xyz[3 * i + 1] = e->ny[i];
      out.xyz[3 * i + 2] = e->nz[i];
      out.state[i] = e->nstate[i];
      out.cid[i] = e->ncid[i];
      int k = out.rowptr[i];
      if (i >= e->edges.head.size()) continue;
      for (int ed = e->edges.head[i]; ed >= 0; ed = e->edges.next[ed]) {
        out.col[k] = e->edges.dst[ed];
        out.w[k] = e->edges.w[ed];
        out.dist[k] = e->edges.dist[ed];
        ++k;
      }
    }
  });
}

void clean_graph(TrgEngine *e) {
  const bool trace = getenv("TRG_TIMING") != nullptr;
  const auto t_cg = Clock::now();
  auto lapc = [&](const char *what) {
    if (trace) fprintf(stderr, "[trg cleanGraph]   %-24s %8.3f ms\n", what, ms_since(t_cg));
  };
  const size_t V = e->nx.size();
  std::vector<int> old2new(V, 0);  // old2new[] default-constructs 0 in the reference too
  std::vector<int> keep_order;     // old ids in the order they receive new ids
  int new_id = 0;
  // new ids follow the iteration order of the reference's unordered_map (trg.cpp:497-504)
  std::vector<int> order;
  node_map_order(e, order);
  for (const int id : order) {
    if (e->nstate[id] == TRG_NODE_INVALID || e->edges.deg[id] < 1) continue;
    old2new[id] = new_id;
    keep_order.push_back(id);
    new_id++;
  }
  // trg.cpp:505-520 drops the edges of kept nodes that lead to a node it deletes.  A kept node has edges,
  // so the only deleted nodes an edge can lead to are Invalid ones: "is deleted" is the state test below
  // (no separate marking pass over the 700 k-entry edge pool).
  lapc("renumbering");
  const int Vn = new_id;
  std::vector<float> x2(Vn), y2(Vn), z2(Vn);
  std::vector<int> st2(Vn), cid2(Vn);
  // the surviving rows in bulk on several host threads: counts, offsets, then every row copied with its
  // targets renumbered (a 6.7 M-entry pool rebuilt push by push cost 65 ms per updateGraph at C3)
  std::vector<int> offs((size_t)Vn + 1, 0);
  parallel_ranges((size_t)Vn, [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; ++k) {
      const int old = keep_order[k];
      int n = 0;
      for (int ed = e->edges.head[old]; ed >= 0; ed = e->edges.next[ed]) n += e->nstate[e->edges.dst[ed]] != TRG_NODE_INVALID;
      offs[k + 1] = n;
    }
  });
  for (int k = 0; k < Vn; ++k) offs[k + 1] += offs[k];
  EdgePool ep;
  ep.alloc_rows(offs);
  parallel_ranges((size_t)Vn, [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; ++k) {
      const int old = keep_order[k];
      x2[k] = e->nx[old];
      y2[k] = e->ny[old];
      z2[k] = e->nz[old];
      st2[k] = e->nstate[old];
      cid2[k] = e->ncid[old];
      int pos = offs[k];
      for (int ed = e->edges.head[old]; ed >= 0; ed = e->edges.next[ed]) {
        const int d = e->edges.dst[ed];
        if (e->nstate[d] == TRG_NODE_INVALID) continue;
        ep.dst[pos] = old2new[d];
        ep.w[pos] = e->edges.w[ed];
        ep.dist[pos] = e->edges.dist[ed];
        ++pos;
      }
    }
    ep.link_rows(offs, k0, k1);
  });
  lapc("rows copied");
  e->last_new2old = keep_order;
  e->nx.swap(x2);
  e->ny.swap(y2);
  e->nz.swap(z2);
  e->nstate.swap(st2);
  e->ncid.swap(cid2);
  e->edges = std::move(ep);
  e->node_id = Vn;
  // new_nodes[new_id] = node for the dense new ids (trg.cpp:502), global_graph.nodes = new_nodes (:526:
  // bucket count, policy and element order of the source are taken over -- what moving it in leaves
  // behind); then the node tree is refilled in that map's iteration order
  if (e->real_map_stale) {
    MapOrderSim new_nodes;
    new_nodes.fill((size_t)Vn);
    e->nodes_sim.assign_from(new_nodes);
    e->nodes_sim.iteration_order(e->kd_insert_order);
  } else {
    std::unordered_map<int, int> new_nodes;
    for (int k = 0; k < Vn; ++k) new_nodes[k] = k;
    e->order_map = std::move(new_nodes);
    e->kd_insert_order.clear();
    for (auto &kv : e->order_map) e->kd_insert_order.push_back(kv.first);
  }
  e->kd_order_dirty = false;
  e->kd_valid = false;
  lapc("container replica");
  grid_rebuild(e);
  lapc("node grid");
  e->host_grid_valid = true;
  e->pool_valid = true;
}

void read_counters(TrgEngine *e) {
  std::vector<DeviceCounters> h(COUNTER_SHARDS);
  if (hipMemcpy(h.data(), e->d_ctr, COUNTER_SHARDS * sizeof(DeviceCounters),
                hipMemcpyDeviceToHost) == hipSuccess) {
    unsigned long long sh = 0, eh = 0, ph = 0, ties = 0;
    for (const DeviceCounters &c : h) {
      sh += c.sample_hits;
      eh += c.edge_hits;
      ph += c.spec_hits;
      ties += c.nn_ties;
    }
    e->stats.bytes_sample_kernel = 12ull * (sh + e->lv_hits_sample);
    e->stats.bytes_edge_kernel = 12ull * eh;
    e->stats.bytes_spec_kernel = 12ull * (ph + e->lv_hits_spec);
    e->stats.map_nn_ties += ties;
  }
}

// ---- local graph (trg.cpp:211-231) ---------------------------------------------------------------
// membership of the local graph (trg.cpp:211-231): n[i] != 0 iff a local-map point lies within
// robot_size / 2 of node i (a disc-emptiness probe); only nodes inside the local map's bounding box
// (grown by that radius) can have one
TrgStatus local_membership(TrgEngine *e, std::vector<int32_t> &n) {
  const size_t V = e->nx.size();
  n.assign(V, 0);
  if (!e->lmap.valid || V == 0) return TRG_OK;
  const float rr = (float)(e->prm.robot_size * 0.5) * 1.01f + 1e-4f;
  const float bx0 = e->lmap.bounds[0] - rr, by0 = e->lmap.bounds[1] - rr;
  const float bx1 = e->lmap.bounds[2] + rr, by1 = e->lmap.bounds[3] + rr;
  std::vector<int> cand;
  std::vector<float> xy;
  for (size_t i = 0; i < V; ++i)
    if (e->nx[i] >= bx0 && e->nx[i] <= bx1 && e->ny[i] >= by0 && e->ny[i] <= by1) {
      cand.push_back((int)i);
      xy.push_back(e->nx[i]);
      xy.push_back(e->ny[i]);
    }
  if (cand.empty()) return TRG_OK;
  std::vector<int32_t> nc(cand.size(), 0);
  TrgStatus st = collision_sync(e, e->lmap, 0.0f, xy.data(), cand.size(), nullptr, nullptr, nc.data(), e->s_aux,
                                (float)(e->prm.robot_size * 0.5));
  if (st != TRG_OK) return st;
  for (size_t k = 0; k < cand.size(); ++k) n[cand[k]] = nc[k];
  return TRG_OK;
}

// member: the membership flags if the caller already has them (updateGraph probes before its host-side
// cleanGraph, while the GPU is still awake: after ~10 ms without work the first launch takes ~2 ms)
TrgStatus set_local_graph(TrgEngine *e, const std::vector<int32_t> *member = nullptr) {
  e->local_nodes.clear();
  e->lkd.clear();
  const size_t V = e->nx.size();
  if (V == 0) {
    e->local_map.clear();
    return TRG_OK;
  }
  std::vector<int32_t> own;
  if (!member || member->size() != V) {
    TrgStatus st = local_membership(e, own);
    if (st != TRG_OK) return st;
    member = &own;
  }
  const std::vector<int32_t> &n = *member;
  e->local_map.clear();  // resetGraph("local"): clear() keeps the bucket array, as the reference's does
  std::vector<int> global_order;
  node_map_order(e, global_order);
  for (int id : global_order) {
    if (n[id] == 0) continue;
    e->local_map[id] = id;
    e->lkd.insert(e->nx[id], e->ny[id], id);
  }
  for (auto &kv : e->local_map) e->local_nodes.push_back(kv.first);
  return TRG_OK;
}

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

// ---- planning (host A*, trg.cpp:537-565, 603-690) ------------------------------------------------
}  // extern "C"

namespace {

// positions in the node tree's insertion order (cleanGraph refills the tree in the node map's iteration
// order, trg.cpp:525-530): only built when an answer really hangs on the tree's shape
void ensure_kd_order_arrays(TrgEngine *e) {
  if (e->kdo_version == e->graph_version) return;
  materialize_kd_order(e);
  const size_t K = e->kd_insert_order.size();
  e->kdo_x.resize(K);
  e->kdo_y.resize(K);
  e->kdo_index.assign(e->nx.size(), -1);
  for (size_t k = 0; k < K; ++k) {
    const int id = e->kd_insert_order[k];
    e->kdo_x[k] = e->nx[id];
    e->kdo_y[k] = e->ny[id];
    e->kdo_index[id] = (int)k;
  }
  e->kdo_version = e->graph_version;
}

// kd_nearest2 on node_tree (trg.cpp:615): the grid answers; an exact fp32 distance tie goes to the
// tree-order argument of kd_tie_winner (no tree is built)
int plan_nearest_node(TrgEngine *e, float qx, float qy, std::vector<int> &tied) {
  bool tie = false;
  int s = e->grid.nearest(qx, qy, &tie);
  if (!tie || s < 0) return s;
  e->stats.nn_ties++;
  e->grid.tied_set(qx, qy, e->grid.dist2(s, qx, qy), tied);
  ensure_kd_order_arrays(e);
  for (int &t : tied) t = e->kdo_index[t];
  std::sort(tied.begin(), tied.end());
  const int w = kd_tie_winner(e->kdo_x.data(), e->kdo_y.data(), (int)e->kdo_x.size(), qx, qy, tied);
  return e->kd_insert_order[w];
}

// First item of kd_nearest_range2(node_tree, goal, robot_size) (trg.cpp:544-546), or -1 when the set is
// empty: the hit set comes from the grid; with several hits the reference takes the one its walk reaches
// LAST (the result list is filled at the head).  A hit within rounding of the radius sends the question
// to the tree replica.
int plan_first_range_hit(TrgEngine *e, float qx, float qy, float r, std::vector<int> &hits) {
  bool doubt = false;
  e->grid.range_set(qx, qy, r, hits, &doubt);
  if (doubt) {
    kd_sync(e);
    e->kd.range(qx, qy, r, hits);
    return hits.empty() ? -1 : hits[0];
  }
  if (hits.empty()) return -1;
  if (hits.size() == 1) return hits[0];
  ensure_kd_order_arrays(e);
  const int K = (int)e->kdo_x.size();
  int last = e->kdo_index[hits[0]];
  for (size_t i = 1; i < hits.size(); ++i) {
    const int h = e->kdo_index[hits[i]];
    if (kd_range_first_of_two(e->kdo_x.data(), e->kdo_y.data(), K, qx, qy, last, h) == last) last = h;
  }
  return e->kd_insert_order[last];
}

// setGoal (trg.cpp:537-565) without side effects: the goal node and whether it lies within robot_size
void plan_goal_node(TrgEngine *e, PlanScratch &ps, const float goal_xyz[3], int *goal, bool *known) {
  const int hit = plan_first_range_hit(e, goal_xyz[0], goal_xyz[1], e->prm.robot_size, ps.hits);
  if (hit >= 0) {
    *goal = hit;
    *known = true;
    return;
  }
  // nearest node by the float norm, first in the node map's iteration order among equals (trg.cpp:549-557)
  float min_dist = std::numeric_limits<float>::max();
  int g = -1;
  std::vector<int> map_order;
  node_map_order(e, map_order);
  for (int id : map_order) {
    const float d = norm2f(e->nx[id] - goal_xyz[0], e->ny[id] - goal_xyz[1]);
    if (d < min_dist) {
      min_dist = d;
      g = id;
    }
  }
  *goal = g;
  *known = false;
}

// planSafePath (trg.cpp:603-690) on the CSR of the global graph: a row's entries are the node's edges in
// the reference's push order, the heap is std::push_heap / std::pop_heap with the reference's comparator
// (f_cost greater-than), so equal-cost ties fall exactly as in the reference's std::priority_queue.
TrgStatus plan_on_csr(TrgEngine *e, PlanScratch &ps, int start, int goal, float *path_xyz, int32_t max_points,
                      TrgPathInfo *info) {
  const Csr &G = e->csr_global;
  const size_t V = e->nx.size();
  const int32_t *rowptr = G.rowptr.data(), *col = G.col.data();
  const float *ew = G.w.data(), *ed = G.dist.data();
  const float *nx = e->nx.data(), *ny = e->ny.data();
  const int *nstate = e->nstate.data();
  ps.begin(V);
  const uint32_t gen = ps.gen;
  std::vector<PlanScratch::Opt> &pool = ps.pool;
  std::vector<int> &heap = ps.heap;
  auto cmp = [&pool](int a, int b) { return pool[a].f > pool[b].f; };
  const float gx = nx[goal], gy = ny[goal];
  const double sf = e->prm.safety_factor;

  info->direct_dist = norm2f(gx - nx[start], gy - ny[start]);
  {
    const double g_cost = 0.0;
    const double f_cost = g_cost + info->direct_dist;
    pool.push_back(PlanScratch::Opt{start, -1, (float)f_cost, (float)g_cost});
    heap.push_back(0);
    ps.open_gen[start] = gen;
    ps.open_idx[start] = 0;
  }
  while (!heap.empty()) {
    std::pop_heap(heap.begin(), heap.end(), cmp);
    const int oi = heap.back();
    heap.pop_back();
    const PlanScratch::Opt cur = pool[oi];
    ps.open_gen[cur.id] = 0;  // open_check.erase(current id)

    if (cur.id == goal) {
      std::vector<int> &chain = ps.chain;
      chain.clear();
      float sum_dist = 0.0, sum_weight = 0.0;
      for (int node = oi; node >= 0; node = pool[node].parent) {
        const PlanScratch::Opt &o = pool[node];
        if (o.parent >= 0) {
          const int pid = pool[o.parent].id;
          for (int k = rowptr[o.id]; k < rowptr[o.id + 1]; ++k)
            if (col[k] == pid) {
              sum_dist += ed[k];
              sum_weight += ew[k];
              break;
            }
        }
        chain.push_back(o.id);
      }
      const float avg_weight = sum_weight / chain.size();
      std::reverse(chain.begin(), chain.end());
      info->path_length = sum_dist;
      info->avg_risk = avg_weight;
      info->num_points = (int32_t)chain.size();
      if (path_xyz) {
        const int m = std::min<int>((int)chain.size(), max_points);
        for (int i = 0; i < m; ++i) {
          path_xyz[3 * i] = nx[chain[i]];
          path_xyz[3 * i + 1] = ny[chain[i]];
          path_xyz[3 * i + 2] = e->nz[chain[i]];
        }
      }
      return TRG_OK;
    }

    ps.close_gen[cur.id] = gen;
    for (int k = rowptr[cur.id]; k < rowptr[cur.id + 1]; ++k) {
      const int dst = col[k];
      if (dst < 0 || dst >= (int)V) continue;
      if (ps.close_gen[dst] == gen || nstate[dst] == TRG_NODE_INVALID) continue;
      const double next_g = cur.g + (sf * ew[k] + 1) * ed[k];
      const double next_f = next_g + norm2f(gx - nx[dst], gy - ny[dst]);
      pool.push_back(PlanScratch::Opt{dst, oi, (float)next_f, (float)next_g});
      const int ni = (int)pool.size() - 1;
      if (ps.open_gen[dst] != gen || pool[ni].g < pool[ps.open_idx[dst]].g) {
        heap.push_back(ni);
        std::push_heap(heap.begin(), heap.end(), cmp);
        ps.open_gen[dst] = gen;
        ps.open_idx[dst] = ni;
      }
    }
  }
  return TRG_ERR_NOT_FOUND;
}

// common head of plan / plan_batch: graph present, CSR rows and node grid current
TrgStatus plan_prepare(TrgEngine *e) {
  const size_t V = e->nx.size();
  if (V == 0) return e->fail(TRG_ERR_NO_GRAPH, "graph is empty");
  const Csr &G = e->csr_global;
  if (G.rowptr.size() != V + 1 || G.state.size() != V) {
    if (!e->pool_valid) return e->fail(TRG_ERR_NO_GRAPH, "no CSR of the current graph");
    snapshot_csr(e, e->csr_global);
  }
  ensure_host_grid(e);
  if (!e->plan_scratch) e->plan_scratch = new PlanScratch();
  return TRG_OK;
}

}  // namespace

extern "C" {

TrgStatus trg_engine_plan(TrgEngine *e, const float start_xy[2], const float goal_xyz[3],
                          float *path_xyz, int32_t max_points, TrgPathInfo *info) {
  if (!e || !start_xy || !goal_xyz || !info) return TRG_ERR_INVALID_ARG;
  info->direct_dist = info->path_length = info->avg_risk = 0.0f;
  info->num_points = 0;
  TrgStatus st = plan_prepare(e);
  if (st != TRG_OK) return st;
  PlanScratch &ps = *e->plan_scratch;
  // setGoal
  e->goal_pose2d[0] = goal_xyz[0];
  e->goal_pose2d[1] = goal_xyz[1];
  plan_goal_node(e, ps, goal_xyz, &e->goal_node, &e->goal_known);
  const int start = plan_nearest_node(e, start_xy[0], start_xy[1], ps.tied);
  st = plan_on_csr(e, ps, start, e->goal_node, path_xyz, max_points, info);
  if (st == TRG_ERR_NOT_FOUND) return e->fail(TRG_ERR_NOT_FOUND, "no path");
  return st;
}

// m consecutive planSafePath calls.  The searches are independent and only read the graph: the start /
// goal nodes are looked up in call order (the goal state the last call leaves is the reference's), the
// searches themselves run on up to 8 host threads, each with its own scratch.
TrgStatus trg_engine_plan_batch(TrgEngine *e, const float *starts_xy, const float *goals_xyz,
                                size_t m, float *path_xyz, int32_t path_cap, int32_t *offsets,
                                TrgPathInfo *infos) {
  if (!e || !offsets || (m && (!starts_xy || !goals_xyz || !infos)))
    return TRG_ERR_INVALID_ARG;
  if (path_cap < 0 || (path_cap > 0 && !path_xyz)) return e->fail(TRG_ERR_INVALID_ARG, "path buffer");
  offsets[0] = 0;
  if (m == 0) return TRG_OK;
  TrgStatus st = plan_prepare(e);
  if (st != TRG_OK) return st;
  PlanScratch &ps0 = *e->plan_scratch;
  std::vector<int> starts(m), goals(m);
  for (size_t k = 0; k < m; ++k) {
    infos[k].direct_dist = infos[k].path_length = infos[k].avg_risk = 0.0f;
    infos[k].num_points = 0;
    e->goal_pose2d[0] = goals_xyz[3 * k];
    e->goal_pose2d[1] = goals_xyz[3 * k + 1];
    plan_goal_node(e, ps0, goals_xyz + 3 * k, &e->goal_node, &e->goal_known);
    goals[k] = e->goal_node;
    starts[k] = plan_nearest_node(e, starts_xy[2 * k], starts_xy[2 * k + 1], ps0.tied);
  }
  std::vector<std::vector<float>> paths(m);
  std::vector<TrgStatus> sts(m, TRG_OK);
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const size_t nthr = std::min<size_t>(std::min<size_t>(m, 8), hw);
  auto work = [&](size_t t, PlanScratch &ps) {
    for (size_t k = t; k < m; k += nthr) {
      TrgPathInfo probe;
      probe.direct_dist = probe.path_length = probe.avg_risk = 0.0f;
      probe.num_points = 0;
      sts[k] = plan_on_csr(e, ps, starts[k], goals[k], nullptr, 0, &probe);
      if (sts[k] == TRG_OK) {
        paths[k].resize(3 * (size_t)probe.num_points);
        for (int i = 0; i < probe.num_points; ++i) {
          const int id = ps.chain[i];
          paths[k][3 * i] = e->nx[id];
          paths[k][3 * i + 1] = e->ny[id];
          paths[k][3 * i + 2] = e->nz[id];
        }
      }
      infos[k] = probe;
    }
  };
  if (nthr <= 1) {
    work(0, ps0);
  } else {
    std::vector<PlanScratch> extra(nthr - 1);
    std::vector<std::thread> thr;
    for (size_t t = 1; t < nthr; ++t) thr.emplace_back(work, t, std::ref(extra[t - 1]));
    work(0, ps0);
    for (auto &th : thr) th.join();
  }
  int32_t used = 0;
  for (size_t k = 0; k < m; ++k) {
    if (sts[k] != TRG_OK && sts[k] != TRG_ERR_NOT_FOUND) return e->fail(sts[k], "plan_batch");
    if (sts[k] == TRG_ERR_NOT_FOUND) infos[k].num_points = 0;
    const int32_t room = path_cap - used;
    const int32_t take = std::min<int32_t>(infos[k].num_points, room);
    if (take > 0) memcpy(path_xyz + 3 * (size_t)used, paths[k].data(), 3 * (size_t)take * sizeof(float));
    used += std::max<int32_t>(take, 0);
    offsets[k + 1] = used;
  }
  return TRG_OK;
}

// reference: TRG::checkReadched (sic) trg.cpp:567-574 and TRG::checkReplan trg.cpp:576-601
int32_t trg_engine_check_reached(TrgEngine *e, const float pos_xy[2]) {
  if (!e || !pos_xy) return 0;
  const float dist = norm2f(e->goal_pose2d[0] - pos_xy[0], e->goal_pose2d[1] - pos_xy[1]);
  return dist < e->prm.goal_tolerance ? 1 : 0;
}

int32_t trg_engine_check_replan(TrgEngine *e, const float pos_xy[2], const float *path_xyz,
                                int32_t n_path) {
  if (!e || !pos_xy) return 0;
  if (e->goal_node < 0 || e->goal_node >= (int)e->nx.size()) return 0;
  const int g = e->goal_node;
  const float dist2subgoal = norm2f(e->nx[g] - pos_xy[0], e->ny[g] - pos_xy[1]);
  if (!e->goal_known && dist2subgoal < e->prm.goal_tolerance) return 1;
  if (!e->goal_known && e->nstate[g] != TRG_NODE_FRONTIER) return 1;
  ensure_host_grid(e);
  std::vector<int> hits;
  for (int i = 0; i < n_path; ++i) {  // (the node grid answers; the tree replica only within rounding of the radius)
    int w = e->grid.within(path_xyz[3 * i], path_xyz[3 * i + 1], e->prm.robot_size);
    if (w < 0) {
      kd_sync(e);
      e->kd.range(path_xyz[3 * i], path_xyz[3 * i + 1], e->prm.robot_size, hits);
      w = hits.empty() ? 0 : 1;
    }
    if (!w) return 1;
  }
  return 0;
}

int32_t trg_engine_refine_path(const float *in_xyz, int32_t n_in, float *out_xyz, int32_t max_out) {
  if (!in_xyz || n_in <= 0) return 0;
  // point_between == 1: p0,p1,p1,p2,p2,...  then a 3-tap mean, last point passed through
  std::vector<float> dense;
  for (int i = 0; i + 1 < n_in; ++i) {
    dense.insert(dense.end(), in_xyz + 3 * i, in_xyz + 3 * i + 3);
    dense.insert(dense.end(), in_xyz + 3 * i + 3, in_xyz + 3 * i + 6);
  }
  const int nd = (int)(dense.size() / 3);
  int written = 0;
  for (int i = 0; i < nd; ++i) {
    float o[3];
    if (i == nd - 1) {
      o[0] = dense[3 * i];
      o[1] = dense[3 * i + 1];
      o[2] = dense[3 * i + 2];
    } else {
      float sum[3] = {0.0f, 0.0f, 0.0f};
      int cnt = 0;
      for (int j = i - 1; j < i + 2; ++j) {
        if (j < 0 || j >= nd) continue;
        for (int k = 0; k < 3; ++k) sum[k] += dense[3 * j + k];
        cnt++;
      }
      for (int k = 0; k < 3; ++k) o[k] = sum[k] / cnt;
    }
    if (out_xyz && written < max_out) {
      out_xyz[3 * written] = o[0];
      out_xyz[3 * written + 1] = o[1];
      out_xyz[3 * written + 2] = o[2];
    }
    written++;
  }
  return written;
}

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
