/*
 * oracle/trg_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's Traversal-Risk-Graph build path, written
 * to follow the reference line by line so that it can act as the parity
 * oracle and as the CPU baseline ("port") of bench.py.  Nothing in the product
 * (trg-planner_amd/) may include, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Reference (all paths relative to /root/reference/cpp/trg_planner/core/trg_planner):
 *   src/graph/trg.cpp      TRG::initGraph :36-64, setGlobalMap :179-193,
 *                          setLocalMap :195-209, setLocalGraph :211-231,
 *                          addNode :233-252, wireEdge :254-370,
 *                          expandGraph :372-454, updateGraph :456-489,
 *                          cleanGraph :491-535, setGoal :537-565,
 *                          planSafePath :603-690, refinePath :692-730,
 *                          isCollision :746-778, isFrontier :780-803
 *   include/graph/trg.h    Edge/NodeState/Node/OptimizeNode :20-48, trgStruct :101-112
 *   src/kdtree/kdtree.c    via oracle/okd.c (own restatement) or, when
 *                          oracle/_ref/libkdtree_ref.so exists, the reference
 *                          kdtree.c itself (set_kd_backend(1)).
 *
 * PARITY PINNING.  The reference ships no tests, golden vectors or fixtures for
 * this path and trg.cpp cannot be compiled here (Eigen, PCL,
 * OpenCV, yaml-cpp absent).  What IS pinned: every spatial query, against the
 * reference kdtree.c compiled in place (tests/test_oracle_kd.py + tests/golden).
 * What is NOT pinned ("parity unpinned"): the Eigen arithmetic inside wireEdge
 * (JacobiSVD of the 3x3 covariance, Eigen un-vendored and unpinned in the
 * reference).  It is restated from Eigen 3.4's published two-sided Jacobi
 * algorithm (Eigen/src/SVD/JacobiSVD.h, Eigen/src/Jacobi/Jacobi.h) in fp32 with
 * plain left-to-right summation; Eigen's own packet summation order for
 * mean/covariance is unknowable here, so weights carry summation-order noise of
 * a few fp32 ulps relative to a real Eigen build.  A second witness
 * (trg_oracle_set_cov_f64) accumulates the same mean / centred rows / covariance
 * in fp64 and rounds once: the HIP engine, which accumulates in fp64 as well,
 * reproduces that witness's weights bit for bit (tests/test_gpu_weight_witness.py,
 * the C3 / C4 full-size digests), while THIS fp32 restatement sits 1-2e-5 away
 * from it on about one near-degenerate edge per million -- the size of the noise
 * any fp32 summation order, Eigen's included, puts on such an edge.
 *
 * Deliberate, documented deviation: the reference seeds std::mt19937 from
 * std::random_device (trg.cpp:20), so it has no canonical sample stream.  The
 * sampler is therefore an explicit input (struct OSampler): mode 0 is the
 * counter-based direction table shared bit-for-bit with the HIP engine, mode 1
 * is the reference's shared mt19937 + uniform_real_distribution<float> stream
 * with a caller-given seed (fidelity experiments only).
 */
// <math.h> first, exactly like the reference's utils/common.h:14: with libstdc++ it injects the
// float overloads into the global namespace, so unqualified cos/sin/atan2/sqrt/fabs on float
// arguments resolve to the fp32 versions, as they do in trg.cpp.
#include <math.h>

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <dlfcn.h>
#include <functional>
#include <limits>
#include <queue>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#include "okd.h"

namespace {

// ---------------------------------------------------------------------------
// kd backend: own restatement (okd.c) or the reference kdtree.c via dlopen.
// ---------------------------------------------------------------------------
struct KdApi {
  void *(*create)();
  void (*clear)(void *);
  void (*free_)(void *);
  int (*insert2)(void *, float, float, void *);
  void *(*nearest2)(void *, float, float);
  void *(*nearest_range2)(void *, float, float, float);
  void (*res_free)(void *);
  int (*res_size)(void *);
  int (*res_end)(void *);
  int (*res_next)(void *);
  void *(*res_item_data)(void *);
};

void *own_create() { return okd_create(); }
KdApi make_own_api() {
  KdApi a;
  a.create = own_create;
  a.clear = [](void *t) { okd_clear((okdtree *)t); };
  a.free_ = [](void *t) { okd_free((okdtree *)t); };
  a.insert2 = [](void *t, float x, float y, void *d) { return okd_insert2((okdtree *)t, x, y, d); };
  a.nearest2 = [](void *t, float x, float y) { return (void *)okd_nearest2((okdtree *)t, x, y); };
  a.nearest_range2 = [](void *t, float x, float y, float r) {
    return (void *)okd_nearest_range2((okdtree *)t, x, y, r);
  };
  a.res_free = [](void *r) { okd_res_free((okdres *)r); };
  a.res_size = [](void *r) { return okd_res_size((okdres *)r); };
  a.res_end = [](void *r) { return okd_res_end((okdres *)r); };
  a.res_next = [](void *r) { return okd_res_next((okdres *)r); };
  a.res_item_data = [](void *r) { return okd_res_item_data((okdres *)r); };
  return a;
}

KdApi g_kd = make_own_api();
int g_kd_backend = 0;
void *g_ref_handle = nullptr;
typedef void *(*ref_create_fn)(int);
ref_create_fn g_ref_create = nullptr;
void *ref_create2() { return g_ref_create(2); }

// ---------------------------------------------------------------------------
// sampler
// ---------------------------------------------------------------------------
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

struct OSampler {
  int mode = 0;  // 0: counter/table, 1: shared mt19937 stream (reference style)
  uint32_t seed = 1;
  int table_bits = 16;
  std::vector<float> cos_t, sin_t;
  std::mt19937 gen;
  std::uniform_real_distribution<float> distr{0.0, 1.0};

  void configure(int m, uint32_t s, int bits) {
    mode = m;
    seed = s;
    table_bits = bits;
    gen.seed(s);
    distr.reset();
    if (mode == 0) build_table();
  }
  // angle = distr * 2 * M_PI (trg.cpp:395) with distr = k / 2^bits (exact in fp32)
  void build_table() {
    size_t n = (size_t)1 << table_bits;
    cos_t.resize(n);
    sin_t.resize(n);
    for (size_t k = 0; k < n; ++k) {
      float u = (float)k / (float)n;
      float angle = u * 2 * M_PI;
      cos_t[k] = cos(angle);
      sin_t[k] = sin(angle);
    }
  }
  // direction of trial `trial` of the expansion of node `id` in build epoch `epoch`
  void direction(uint32_t epoch, uint32_t id, uint32_t trial, float &c, float &s) {
    if (mode == 0) {
      uint32_t k = sample_hash(seed, epoch, id, trial) >> (32 - table_bits);
      c = cos_t[k];
      s = sin_t[k];
    } else {
      float angle = distr(gen) * 2 * M_PI;  // trg.cpp:395
      c = cos(angle);
      s = sin(angle);
    }
  }
  // uniform [0,1) for the root retries (trg.cpp:53-54); counter k = 2*cnt, 2*cnt+1
  float uniform(uint32_t epoch, uint32_t k) {
    if (mode == 0) {
      uint32_t h = sample_hash(seed, epoch, 0xFFFFFFFFu, k);
      return (float)(h >> 8) * (1.0f / 16777216.0f);
    }
    return distr(gen);
  }
};

// ---------------------------------------------------------------------------
// Eigen restatement: JacobiSVD<MatrixXf>(cov, ComputeFullU) on a 3x3 matrix.
// Follows Eigen 3.4 Eigen/src/SVD/JacobiSVD.h (compute(): scaling, sweep order
// p=1..n-1 / q=0..p-1, threshold = max(min, 2*eps*maxDiag), sign fix, sort) and
// Eigen/src/Jacobi/Jacobi.h (makeJacobi, rotation product, rotation in the plane).
// Column-major 3x3 stored as m[r][c].
// ---------------------------------------------------------------------------
struct Rot {
  float c, s;
};
inline Rot rot_mul(Rot a, Rot b) { return Rot{a.c * b.c - a.s * b.s, a.c * b.s + a.s * b.c}; }
inline Rot rot_T(Rot a) { return Rot{a.c, -a.s}; }

// rows p,q of M: x' = c x + s y ; y' = -s x + c y   (applyOnTheLeft)
inline void rot_left(float M[3][3], int p, int q, Rot j) {
  if (j.c == 1.0f && j.s == 0.0f) return;
  for (int i = 0; i < 3; ++i) {
    float xi = M[p][i], yi = M[q][i];
    M[p][i] = j.c * xi + j.s * yi;
    M[q][i] = -j.s * xi + j.c * yi;
  }
}
// cols p,q of M with j.transpose()               (applyOnTheRight)
inline void rot_right(float M[3][3], int p, int q, Rot j) {
  Rot t = rot_T(j);
  if (t.c == 1.0f && t.s == 0.0f) return;
  for (int i = 0; i < 3; ++i) {
    float xi = M[i][p], yi = M[i][q];
    M[i][p] = t.c * xi + t.s * yi;
    M[i][q] = -t.s * xi + t.c * yi;
  }
}

inline bool make_jacobi(float x, float y, float z, Rot &r) {
  float deno = 2.0f * std::fabs(y);
  if (deno < std::numeric_limits<float>::min()) {
    r.c = 1.0f;
    r.s = 0.0f;
    return false;
  }
  float tau = (x - z) / deno;
  float w = std::sqrt(tau * tau + 1.0f);
  float t;
  if (tau > 0.0f) {
    t = 1.0f / (tau + w);
  } else {
    t = 1.0f / (tau - w);
  }
  float sign_t = t > 0.0f ? 1.0f : -1.0f;
  float n = 1.0f / std::sqrt(t * t + 1.0f);
  r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
  r.c = n;
  return true;
}

inline void real_2x2_jacobi_svd(float W[3][3], int p, int q, Rot &j_left, Rot &j_right) {
  float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
  Rot rot1;
  float t = m00 + m11;
  float d = m10 - m01;
  if (std::fabs(d) < std::numeric_limits<float>::min()) {
    rot1.s = 0.0f;
    rot1.c = 1.0f;
  } else {
    float u = t / d;
    float tmp = std::sqrt(1.0f + u * u);
    rot1.s = 1.0f / tmp;
    rot1.c = u / tmp;
  }
  // m.applyOnTheLeft(0,1,rot1)
  if (!(rot1.c == 1.0f && rot1.s == 0.0f)) {
    float a0 = m00, b0 = m10, a1 = m01, b1 = m11;
    m00 = rot1.c * a0 + rot1.s * b0;
    m10 = -rot1.s * a0 + rot1.c * b0;
    m01 = rot1.c * a1 + rot1.s * b1;
    m11 = -rot1.s * a1 + rot1.c * b1;
  }
  make_jacobi(m00, m01, m11, j_right);
  j_left = rot_mul(rot1, rot_T(j_right));
}

// U (3x3) of the SVD of A; returns U in U[r][c], columns sorted by decreasing singular value
void jacobi_svd_u3(const float A[3][3], float U[3][3]) {
  float W[3][3];
  float scale = 0.0f;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) scale = std::max(scale, std::fabs(A[r][c]));
  if (scale == 0.0f) scale = 1.0f;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      W[r][c] = A[r][c] / scale;
      U[r][c] = (r == c) ? 1.0f : 0.0f;
    }
  const float precision = 2.0f * std::numeric_limits<float>::epsilon();
  const float considerAsZero = std::numeric_limits<float>::min();
  float maxDiag = std::max(std::fabs(W[0][0]), std::max(std::fabs(W[1][1]), std::fabs(W[2][2])));
  bool finished = false;
  int guard = 0;
  while (!finished && guard++ < 1000) {
    finished = true;
    for (int p = 1; p < 3; ++p) {
      for (int q = 0; q < p; ++q) {
        float threshold = std::max(considerAsZero, precision * maxDiag);
        if (std::fabs(W[p][q]) > threshold || std::fabs(W[q][p]) > threshold) {
          finished = false;
          Rot jl, jr;
          real_2x2_jacobi_svd(W, p, q, jl, jr);
          rot_left(W, p, q, jl);
          rot_right(U, p, q, rot_T(jl));
          rot_right(W, p, q, jr);
          maxDiag = std::max(maxDiag, std::max(std::fabs(W[p][p]), std::fabs(W[q][q])));
        }
      }
    }
  }
  float sv[3];
  for (int i = 0; i < 3; ++i) {
    float a = std::fabs(W[i][i]);
    sv[i] = a;
    if (a != 0.0f) {
      float sgn = W[i][i] / a;
      for (int r = 0; r < 3; ++r) U[r][i] *= sgn;
    }
  }
  for (int i = 0; i < 3; ++i) sv[i] *= scale;
  for (int i = 0; i < 3; ++i) {
    int pos = 0;
    float mx = sv[i];
    for (int k = i; k < 3; ++k)
      if (sv[k] > mx) {
        mx = sv[k];
        pos = k - i;
      }
    if (mx == 0.0f) break;
    if (pos) {
      pos += i;
      std::swap(sv[i], sv[pos]);
      for (int r = 0; r < 3; ++r) std::swap(U[r][i], U[r][pos]);
    }
  }
}

// ---------------------------------------------------------------------------
// graph types (trg.h:20-48, 101-112)
// ---------------------------------------------------------------------------
struct Pt {
  float x, y, z, pad;  // pcl::PointXYZ is 16 bytes
};

struct OEdge {
  OEdge(int d, float w, float l) : dst_id_(d), weight_(w), dist_(l) {}
  int dst_id_;
  float weight_;
  float dist_;
};

enum : int { ST_VALID = 0, ST_INVALID = -1, ST_FRONTIER = 1 };

struct ONode {
  ONode(int id, float x, float y, float z, int st) : id_(id), state_(st) {
    pos_[0] = x;
    pos_[1] = y;
    pos_[2] = z;
  }
  int id_;
  float pos_[3];
  int state_;
  std::vector<OEdge *> edges_;
  int cid_ = -1;  // creation id inside the current build (instrumentation)
};

struct OptimizeNode {
  OptimizeNode(int i, float f, float g) : id_(i), f_(f), g_(g) {}
  int id_;
  OptimizeNode *parent_;
  float f_;
  float g_;
};

struct OGraph {
  std::unordered_map<int, ONode *> nodes;
  void *node_tree = nullptr;
  int node_id = 0;
  float root_pos[2] = {0, 0};
  void *map_tree = nullptr;
  std::vector<Pt> cloud;
};

struct OParams {
  float expand_dist, robot_size;
  int sample_num;
  float height_threshold, collision_threshold, update_collision_threshold, safety_factor,
      goal_tolerance;
};

struct OCounters {
  uint64_t collision_queries = 0, collision_hits = 0;
  uint64_t nn_map_queries = 0;
  uint64_t ellipse_queries = 0, ellipse_hits = 0;
  uint64_t wire_calls = 0, wire_evals = 0, wire_ok = 0;
  uint64_t wire_gate = 0, wire_seg = 0, wire_empty = 0, wire_few = 0, wire_clamped = 0;
  uint64_t expanded = 0, trials = 0, samples = 0, created = 0, invalid_created = 0;
  uint64_t nn_node_queries = 0;
  uint64_t sample_hits = 0;  // the part of collision_hits issued by expandGraph's sampling loop (trg.cpp:398)
  // map points inside the queries (segment discs + ellipse gather) of wireEdge(node, new_node)
  // (trg.cpp:425) and of every other wireEdge call
  uint64_t wire_hits_new = 0, wire_hits_other = 0;
};

inline float norm2(float dx, float dy) { return std::sqrt(dx * dx + dy * dy); }

struct WireTrace {  // one record per wireEdge() call that got past the dedupe
  int src_cid, dst_cid;
  int status;  // 0 ok, 1 gate, 2 segment collision, 3 empty gather, 4 <3 points
  int n_pts;
  float weight, dist;
};

class Oracle {
 public:
  explicit Oracle(const OParams &p) : param_(p), kd_(g_kd) {
    for (OGraph *g : {&global_, &local_}) {
      g->node_tree = kd_.create();
      g->map_tree = kd_.create();
    }
  }
  ~Oracle() {
    for (OGraph *g : {&global_, &local_}) {
      kd_.free_(g->node_tree);
      kd_.free_(g->map_tree);
    }
  }

  OGraph &graph(int type) { return type == 0 ? global_ : local_; }

  // trg.cpp:732-744
  void resetGraph(int type) {
    OGraph &g = graph(type);
    g.nodes.clear();
    kd_.clear(g.node_tree);
    g.node_id = 0;
  }
  void resetMap(int type) {
    OGraph &g = graph(type);
    kd_.clear(g.map_tree);
    g.cloud.clear();
  }

  // trg.cpp:179-193
  void setGlobalMap(const float *xyz, size_t n, size_t stride) {
    OGraph &g = global_;
    resetMap(0);
    g.cloud.resize(n);
    for (size_t i = 0; i < n; ++i) {
      g.cloud[i] = Pt{xyz[i * stride], xyz[i * stride + 1], xyz[i * stride + 2], 1.0f};
    }
    for (size_t i = 0; i < n; ++i) {
      Pt &pt = g.cloud[i];
      kd_.insert2(g.map_tree, pt.x, pt.y, &pt);
    }
  }

  // trg.cpp:195-209
  void setLocalMap(float sx, float sy, const float *xyz, size_t n, size_t stride) {
    OGraph &g = local_;
    resetMap(1);
    g.root_pos[0] = sx;
    g.root_pos[1] = sy;
    g.cloud.resize(n);
    for (size_t i = 0; i < n; ++i) {
      g.cloud[i] = Pt{xyz[i * stride], xyz[i * stride + 1], xyz[i * stride + 2], 1.0f};
    }
    for (size_t i = 0; i < n; ++i) {
      Pt &pt = g.cloud[i];
      kd_.insert2(g.map_tree, pt.x, pt.y, &pt);
    }
    setLocalGraph();
  }

  // trg.cpp:211-231
  void setLocalGraph() {
    resetGraph(1);
    for (auto &node : global_.nodes) {
      void *res = kd_.nearest_range2(local_.map_tree, node.second->pos_[0], node.second->pos_[1],
                                      param_.robot_size * 0.5);
      if (kd_.res_size(res) == 0) {
        kd_.res_free(res);
        continue;
      }
      local_.nodes[node.first] = node.second;
      kd_.insert2(local_.node_tree, node.second->pos_[0], node.second->pos_[1], node.second);
      kd_.res_free(res);
    }
  }

  // trg.cpp:746-778
  bool isCollision(float px, float py, int type, float threshold, int *out_cnt = nullptr,
                   int *out_n = nullptr) {
    OGraph &g = graph(type);
    void *res = kd_.nearest_range2(g.map_tree, px, py, param_.robot_size);
    cnt_.collision_queries++;
    if (kd_.res_size(res) == 0) {
      kd_.res_free(res);
      if (out_cnt) *out_cnt = 0;
      if (out_n) *out_n = 0;
      return true;
    }
    std::vector<Pt *> pts;
    float z_med = 0.0;
    while (!kd_.res_end(res)) {
      Pt *pt = reinterpret_cast<Pt *>(kd_.res_item_data(res));
      pts.push_back(pt);
      kd_.res_next(res);
    }
    kd_.res_free(res);
    cnt_.collision_hits += pts.size();

    std::sort(pts.begin(), pts.end(), [](Pt *a, Pt *b) { return a->z < b->z; });
    z_med = pts[pts.size() / 2]->z;

    int total = pts.size();
    int cnt = 0;
    for (auto &pt : pts) {
      if (fabs(pt->z - z_med) > param_.height_threshold) {
        cnt++;
      }
    }
    if (out_cnt) *out_cnt = cnt;
    if (out_n) *out_n = total;
    float ratio = static_cast<float>(cnt) / total;
    if (ratio > threshold) {
      return true;
    }
    return false;
  }

  // nearest map point in 2-D -> its z (trg.cpp:244-247); *tie set if another map point has the
  // identical fp32 squared distance (the winner then depends on tree shape)
  float nearestZ(float px, float py, int type, bool *ok = nullptr) {
    OGraph &g = graph(type);
    void *res = kd_.nearest2(g.map_tree, px, py);
    cnt_.nn_map_queries++;
    if (!res) {
      if (ok) *ok = false;
      return 0.0f;
    }
    Pt *pt = reinterpret_cast<Pt *>(kd_.res_item_data(res));
    kd_.res_free(res);
    if (ok) *ok = true;
    return pt->z;
  }

  // trg.cpp:233-252
  bool addNode(int node_id, float px, float py, int state, int type) {
    OGraph &g = graph(type);
    if (node_id == 0) {
      if (isCollision(px, py, type, param_.collision_threshold)) {
        return false;
      }
    }
    float z = nearestZ(px, py, type);
    ONode *node = new ONode(node_id, px, py, z, state);
    node->cid_ = next_cid_++;
    all_created_.push_back(node);
    g.nodes[node_id] = node;
    kd_.insert2(g.node_tree, px, py, node);
    g.node_id++;
    cnt_.created++;
    return true;
  }

  // The pure part of wireEdge (trg.cpp:269-363): everything that depends only on the two
  // endpoint positions and the map.  status: 0 ok, 1 slope gate, 2 segment collision,
  // 3 empty gather, 4 fewer than 3 points.
  int edgeRisk(const float p1[3], const float p2[3], int type, float &weight, float &dist_out,
               int &n_pts) {
    weight = 0.0f;
    n_pts = 0;
    float max_slope = atan2(param_.height_threshold, param_.robot_size);
    float slope = atan2(fabs(p1[2] - p2[2]), norm2(p1[0] - p2[0], p1[1] - p2[1]));
    float dist = norm2(p1[0] - p2[0], p1[1] - p2[1]);
    dist_out = dist;
    if (slope > max_slope) {
      return 1;
    }
    // dir = (p2 - p1).normalized()  (Eigen: divide by sqrt(squaredNorm) when > 0)
    float ddx = p2[0] - p1[0], ddy = p2[1] - p1[1];
    float sq = ddx * ddx + ddy * ddy;
    float dirx = ddx, diry = ddy;
    if (sq > 0.0f) {
      float nrm = std::sqrt(sq);
      dirx = ddx / nrm;
      diry = ddy / nrm;
    }
    // center = p1 + 0.5 * dist * dir   (0.5*dist is exact; Eigen narrows the scalar to float)
    float half = (float)(0.5 * dist);
    float cx = p1[0] + half * dirx;
    float cy = p1[1] + half * diry;

    float ds = param_.robot_size * 0.5;
    for (float i = 0; i < dist; i += ds) {
      float qx = p1[0] + i * dirx;
      float qy = p1[1] + i * diry;
      if (isCollision(qx, qy, type, param_.collision_threshold)) {
        return 2;
      }
    }

    float c = 0.5 * dist;
    float b = param_.robot_size;
    float a = b;
    if (c >= b) {
      a = sqrt(c * c + b * b);
    }
    bool isCircle = (a == b) ? true : false;

    OGraph &g = graph(type);
    // R << dir.x, -dir.y, dir.y, dir.x   (trg.cpp:302-303)
    float r00 = dirx, r01 = -diry, r10 = diry, r11 = dirx;
    void *res = kd_.nearest_range2(g.map_tree, cx, cy, a);
    cnt_.ellipse_queries++;
    cnt_.ellipse_hits += kd_.res_size(res);
    if (kd_.res_size(res) == 0) {
      kd_.res_free(res);
      return 3;
    }
    std::vector<Pt> ell;
    while (!kd_.res_end(res)) {
      Pt *pt = reinterpret_cast<Pt *>(kd_.res_item_data(res));
      float qx = pt->x - cx, qy = pt->y - cy;
      Pt p;
      p.x = r00 * qx + r01 * qy;
      p.y = r10 * qx + r11 * qy;
      p.z = pt->z;
      p.pad = 1.0f;
      if (isCircle) {
        ell.push_back(p);
      } else {
        if ((p.x * p.x) * (b * b) + (p.y * p.y) * (a * a) < a * a * b * b) {
          ell.push_back(p);
        }
      }
      kd_.res_next(res);
    }
    kd_.res_free(res);
    n_pts = (int)ell.size();
    if (ell.size() < 3) {
      return 4;
    }

    // mean / centred / covariance (trg.cpp:332-338)
    int n = (int)ell.size();
    float cov[3][3];
    if (!cov_f64_) {
      float mean[3] = {0, 0, 0};
      for (int i = 0; i < n; ++i) {
        mean[0] += ell[i].x;
        mean[1] += ell[i].y;
        mean[2] += ell[i].z;
      }
      for (int k = 0; k < 3; ++k) mean[k] = mean[k] / (float)n;
      float acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int i = 0; i < n; ++i) {
        float v[3] = {ell[i].x - mean[0], ell[i].y - mean[1], ell[i].z - mean[2]};
        for (int r = 0; r < 3; ++r)
          for (int cc = 0; cc < 3; ++cc) acc[r][cc] += v[r] * v[cc];
      }
      float denom = (float)static_cast<double>(n - 1);
      for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) cov[r][cc] = acc[r][cc] / denom;
    } else {
      // experiment switch: same formula with fp64 accumulation, rounded to fp32 once
      double mean[3] = {0, 0, 0};
      for (int i = 0; i < n; ++i) {
        mean[0] += ell[i].x;
        mean[1] += ell[i].y;
        mean[2] += ell[i].z;
      }
      for (int k = 0; k < 3; ++k) mean[k] /= (double)n;
      double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int i = 0; i < n; ++i) {
        double v[3] = {ell[i].x - mean[0], ell[i].y - mean[1], ell[i].z - mean[2]};
        for (int r = 0; r < 3; ++r)
          for (int cc = 0; cc < 3; ++cc) acc[r][cc] += v[r] * v[cc];
      }
      for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) cov[r][cc] = (float)(acc[r][cc] / (double)(n - 1));
    }

    float U[3][3];
    jacobi_svd_u3(cov, U);
    // eigenvectors = U.normalized(): divide by the Frobenius norm (trg.cpp:340)
    float fro = 0.0f;
    for (int cc = 0; cc < 3; ++cc)
      for (int r = 0; r < 3; ++r) fro += U[r][cc] * U[r][cc];
    if (fro > 0.0f) {
      float nrm = std::sqrt(fro);
      for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) U[r][cc] = U[r][cc] / nrm;
    }
    // col(k).dot(-gravity), gravity = (0,0,-1)  (trg.cpp:347-354)
    float hor_grad = U[0][0] * 0.0f + U[1][0] * 0.0f + U[2][0] * 1.0f;
    float ver_grad = U[0][1] * 0.0f + U[1][1] * 0.0f + U[2][1] * 1.0f;
    if (hor_grad < 0) {
      hor_grad = (-U[0][0]) * 0.0f + (-U[1][0]) * 0.0f + (-U[2][0]) * 1.0f;
    }
    if (ver_grad < 0) {
      ver_grad = (-U[0][1]) * 0.0f + (-U[1][1]) * 0.0f + (-U[2][1]) * 1.0f;
    }
    float ratio = 0.8;
    float w = ratio * hor_grad + (1 - ratio) * ver_grad;
    if (w < 0.1) {
      w = 0.0;
      cnt_.wire_clamped++;
    }
    weight = w;
    return 0;
  }

  // trg.cpp:254-370
  void wireEdge(ONode *node1, ONode *node2, int type) {
    cnt_.wire_calls++;
    if (node1->id_ == node2->id_) {
      return;
    }
    for (auto &edge : node1->edges_) {
      if (edge->dst_id_ == node2->id_) {
        return;
      }
    }
    for (auto &edge : node2->edges_) {
      if (edge->dst_id_ == node1->id_) {
        return;
      }
    }
    cnt_.wire_evals++;
    float weight, dist;
    int n_pts;
    const uint64_t hits_before = cnt_.collision_hits + cnt_.ellipse_hits;
    int status = edgeRisk(node1->pos_, node2->pos_, type, weight, dist, n_pts);
    cnt_.wire_hits_other += cnt_.collision_hits + cnt_.ellipse_hits - hits_before;
    if (trace_wires_) {
      wire_trace_.push_back(WireTrace{node1->cid_, node2->cid_, status, n_pts, weight, dist});
    }
    switch (status) {
      case 1: cnt_.wire_gate++; return;
      case 2: cnt_.wire_seg++; return;
      case 3: cnt_.wire_empty++; return;
      case 4: cnt_.wire_few++; return;
      default: break;
    }
    cnt_.wire_ok++;
    OEdge *edge_1 = new OEdge(node2->id_, weight, dist);
    OEdge *edge_2 = new OEdge(node1->id_, weight, dist);
    node1->edges_.push_back(edge_1);
    node2->edges_.push_back(edge_2);
  }

  // trg.cpp:372-454
  void expandGraph(int ref_id, int type) {
    OGraph &g = graph(type);
    ONode *ref_node = g.nodes.at(ref_id);

    std::deque<ONode *> expand_queue;
    expand_queue.push_back(ref_node);
    while (!expand_queue.empty()) {
      ONode *node = expand_queue.front();
      expand_queue.pop_front();
      cnt_.expanded++;

      std::vector<std::pair<float, float>> samples;
      int max_trial_sample = 1000;
      int trial_sample = 0;
      uint32_t draw = 0;
      while ((int)samples.size() < param_.sample_num) {
        if (trial_sample > max_trial_sample) {
          break;
        }
        float expand_dist = param_.expand_dist;
        float cs, sn;
        sampler_.direction(epoch_, (uint32_t)node->id_, draw++, cs, sn);
        cnt_.trials++;
        float sx = node->pos_[0] + expand_dist * cs;
        float sy = node->pos_[1] + expand_dist * sn;
        const uint64_t hits_before = cnt_.collision_hits;
        const bool sample_collides = isCollision(sx, sy, type, param_.collision_threshold);
        cnt_.sample_hits += cnt_.collision_hits - hits_before;
        if (sample_collides || !inCore(sx, sy)) {
          trial_sample++;
          continue;
        }
        samples.push_back({sx, sy});
      }
      cnt_.samples += samples.size();

      for (auto &sample : samples) {
        void *res = kd_.nearest2(g.node_tree, sample.first, sample.second);
        ONode *existing_node = reinterpret_cast<ONode *>(kd_.res_item_data(res));
        kd_.res_free(res);
        cnt_.nn_node_queries++;
        if (existing_node->state_ == ST_INVALID) {
          continue;
        }
        if (norm2(existing_node->pos_[0] - sample.first, existing_node->pos_[1] - sample.second) <
            param_.robot_size) {
          wireEdge(node, existing_node, type);
          continue;
        }

        int new_state = (ref_id == 0) ? ST_VALID : ST_FRONTIER;
        if (!addNode(g.node_id, sample.first, sample.second, new_state, type)) {
          continue;
        }
        ONode *new_node = g.nodes.at(g.node_id - 1);
        {
          const uint64_t before = cnt_.collision_hits + cnt_.ellipse_hits, other = cnt_.wire_hits_other;
          wireEdge(node, new_node, type);
          cnt_.wire_hits_new += cnt_.collision_hits + cnt_.ellipse_hits - before;
          cnt_.wire_hits_other = other;  // (wireEdge books every call as "other"; this one is not)
        }

        if (param_.expand_dist - param_.robot_size < 0.25 * param_.expand_dist) {
          void *res2 = kd_.nearest_range2(g.node_tree, new_node->pos_[0], new_node->pos_[1],
                                           param_.expand_dist);
          if (kd_.res_size(res2) > 0) {
            while (!kd_.res_end(res2)) {
              ONode *ex = reinterpret_cast<ONode *>(kd_.res_item_data(res2));
              if (ex->state_ == ST_INVALID) {
                kd_.res_next(res2);
                continue;
              }
              wireEdge(new_node, ex, type);
              kd_.res_next(res2);
            }
          }
          kd_.res_free(res2);
        }

        if (new_node->edges_.size() < 1) {
          new_node->state_ = ST_INVALID;
          cnt_.invalid_created++;
          continue;
        }
        expand_queue.push_back(new_node);
      }
    }
  }

  // trg.cpp:36-64 ; returns false where the reference would exit(1)
  bool initGraph(const float start3d[3]) {
    OGraph &g = global_;
    resetGraph(0);
    epoch_ = epoch_base_;
    next_cid_ = 0;
    all_created_.clear();
    wire_trace_.clear();
    if (g.cloud.empty()) return false;

    g.root_pos[0] = start3d[0];
    g.root_pos[1] = start3d[1];
    float rx = g.root_pos[0], ry = g.root_pos[1];
    rx = rx + param_.expand_dist;
    int cnt = 0;
    while (!inCore(rx, ry) || !addNode(g.node_id, rx, ry, ST_VALID, 0)) {
      if (cnt > 100) {
        return false;
      }
      float u0 = sampler_.uniform(epoch_, 2 * cnt);
      float u1 = sampler_.uniform(epoch_, 2 * cnt + 1);
      rx = rx + param_.expand_dist * u0;
      ry = ry + param_.expand_dist * u1;
      cnt++;
    }
    expandGraph(g.node_id - 1, 0);
    snapshotPreClean();
    cleanGraph(false);
    return true;
  }

  // trg.cpp:491-535
  void cleanGraph(bool updateLocal) {
    OGraph &g = global_;
    std::unordered_map<int, int> old2new;
    std::unordered_map<int, ONode *> new_nodes;
    std::vector<int> del_edges;
    int new_id = 0;
    for (auto &node : g.nodes) {
      if (node.second->state_ == ST_INVALID || node.second->edges_.size() < 1) {
        continue;
      }
      node.second->id_ = new_id;
      new_nodes[new_id] = node.second;
      old2new[node.first] = new_id;
      new_id++;
      for (auto &edge : node.second->edges_) {
        if (g.nodes[edge->dst_id_]->state_ == ST_INVALID) {
          del_edges.push_back(edge->dst_id_);
        }
      }
    }
    // the reference does a linear std::find over del_edges (trg.cpp:515); a sorted copy
    // gives the identical membership answer without the O(E*|del|) blow-up
    std::vector<int> del_sorted(del_edges);
    std::sort(del_sorted.begin(), del_sorted.end());
    for (auto &node : new_nodes) {
      std::vector<OEdge *> new_edges;
      for (auto &edge : node.second->edges_) {
        if (std::binary_search(del_sorted.begin(), del_sorted.end(), edge->dst_id_)) {
          continue;
        }
        OEdge *new_edge = new OEdge(old2new[edge->dst_id_], edge->weight_, edge->dist_);
        new_edges.push_back(new_edge);
      }
      node.second->edges_.clear();
      node.second->edges_ = new_edges;
    }

    resetGraph(0);
    g.nodes = new_nodes;
    g.node_id = new_id;
    for (auto &node : g.nodes) {
      kd_.insert2(g.node_tree, node.second->pos_[0], node.second->pos_[1], node.second);
    }
    if (updateLocal) {
      setLocalGraph();
    }
  }

  // trg.cpp:780-803
  bool isFrontier(float px, float py) {
    float dx = px - local_.root_pos[0], dy = py - local_.root_pos[1];
    float sq = dx * dx + dy * dy;
    if (sq > 0.0f) {
      float nrm = std::sqrt(sq);
      dx = dx / nrm;
      dy = dy / nrm;
    }
    float k = 2 * param_.robot_size;
    float chx = px + k * dx, chy = py + k * dy;
    void *res2 = kd_.nearest_range2(global_.node_tree, chx, chy, param_.robot_size);
    if (kd_.res_size(res2) > 0) {
      kd_.res_free(res2);
      return false;
    }
    kd_.res_free(res2);
    void *res1 = kd_.nearest_range2(local_.map_tree, chx, chy, 0.5 * param_.robot_size);
    if (kd_.res_size(res1) == 0) {
      kd_.res_free(res1);
      return true;
    }
    kd_.res_free(res1);
    return false;
  }

  // trg.cpp:456-489
  void updateGraph() {
    epoch_++;
    std::deque<ONode *> expand_queue;
    for (auto &node : local_.nodes) {
      float nx = node.second->pos_[0], ny = node.second->pos_[1];
      if (norm2(nx - local_.root_pos[0], ny - local_.root_pos[1]) > 2.0 * param_.expand_dist) {
        if (isCollision(nx, ny, 1, param_.update_collision_threshold) ||
            node.second->edges_.size() < 1) {
          node.second->state_ = ST_INVALID;
          continue;
        }
      }
      if (isFrontier(nx, ny) && node.second->state_ == ST_FRONTIER) {
        node.second->state_ = ST_FRONTIER;
        expand_queue.push_back(node.second);
        continue;
      }
      expand_queue.push_back(node.second);
      node.second->state_ = ST_VALID;
    }
    while (!expand_queue.empty()) {
      ONode *node = expand_queue.front();
      expand_queue.pop_front();
      expandGraph(node->id_, 0);
    }
    cleanGraph(true);
  }

  // trg.cpp:537-565
  void setGoal(const float goal[3]) {
    OGraph &g = global_;
    goal_pose2d_[0] = goal[0];
    goal_pose2d_[1] = goal[1];
    void *res = kd_.nearest_range2(g.node_tree, goal[0], goal[1], param_.robot_size);
    if (kd_.res_size(res) == 0) {
      kd_.res_free(res);
      float min_dist = std::numeric_limits<float>::max();
      for (auto &node : g.nodes) {
        float dist = norm2(node.second->pos_[0] - goal[0], node.second->pos_[1] - goal[1]);
        if (dist < min_dist) {
          min_dist = dist;
          goal_node_ = node.second;
        }
      }
      goal_known_ = false;
    } else {
      goal_node_ = reinterpret_cast<ONode *>(kd_.res_item_data(res));
      kd_.res_free(res);
      goal_known_ = true;
    }
  }

  // trg.cpp:603-690
  bool planSafePath(const float start2d[2], const float goal[3], std::vector<float> &out_path,
                    float &direct_dist, float &path_length, float &avg_risk) {
    setGoal(goal);
    OGraph &g = global_;
    void *res = kd_.nearest2(g.node_tree, start2d[0], start2d[1]);
    ONode *start_node = reinterpret_cast<ONode *>(kd_.res_item_data(res));
    kd_.res_free(res);

    std::priority_queue<OptimizeNode *, std::vector<OptimizeNode *>,
                        std::function<bool(OptimizeNode *, OptimizeNode *)>>
        open_list([](OptimizeNode *a, OptimizeNode *b) { return a->f_ > b->f_; });
    std::vector<OptimizeNode *> open_check(g.nodes.size(), nullptr);
    std::vector<OptimizeNode *> pool;

    direct_dist = norm2(goal_node_->pos_[0] - start_node->pos_[0],
                        goal_node_->pos_[1] - start_node->pos_[1]);
    double g_cost = 0.0;
    double f_cost = g_cost + direct_dist;
    OptimizeNode *st = new OptimizeNode(start_node->id_, f_cost, g_cost);
    pool.push_back(st);
    st->parent_ = nullptr;
    open_list.push(st);
    open_check[st->id_] = st;
    std::vector<OptimizeNode *> close_list(g.nodes.size(), nullptr);
    bool found = false;
    std::vector<std::array<float, 3>> path;

    while (!open_list.empty()) {
      OptimizeNode *opti_node = open_list.top();
      open_list.pop();
      open_check[opti_node->id_] = nullptr;

      if (opti_node->id_ == goal_node_->id_) {
        OptimizeNode *node = opti_node;
        float sum_dist = 0.0;
        float sum_weight = 0.0;
        float avg_weight = 0.0;
        while (node != nullptr) {
          ONode *n = g.nodes.at(node->id_);
          for (auto &edge : n->edges_) {
            if (node->parent_ != nullptr && edge->dst_id_ == node->parent_->id_) {
              sum_dist += edge->dist_;
              sum_weight += edge->weight_;
              break;
            }
          }
          path.push_back({n->pos_[0], n->pos_[1], n->pos_[2]});
          node = node->parent_;
        }
        avg_weight = sum_weight / path.size();
        std::reverse(path.begin(), path.end());
        path_length = sum_dist;
        avg_risk = avg_weight;
        found = true;
        break;
      }

      ONode *curr_node = g.nodes.at(opti_node->id_);
      close_list[curr_node->id_] = opti_node;

      for (auto e : curr_node->edges_) {
        ONode *dst_node = g.nodes.at(e->dst_id_);
        if (close_list[dst_node->id_] != nullptr || dst_node->state_ == ST_INVALID) {
          continue;
        }
        double next_g_cost = opti_node->g_ + (param_.safety_factor * e->weight_ + 1) * e->dist_;
        double next_f_cost = next_g_cost + norm2(goal_node_->pos_[0] - dst_node->pos_[0],
                                                 goal_node_->pos_[1] - dst_node->pos_[1]);
        OptimizeNode *dst = new OptimizeNode(dst_node->id_, next_f_cost, next_g_cost);
        pool.push_back(dst);
        dst->parent_ = opti_node;
        if (open_check[dst_node->id_] == nullptr) {
          open_list.push(dst);
          open_check[dst_node->id_] = dst;
        } else if (dst->g_ < open_check[dst_node->id_]->g_) {
          open_list.push(dst);
          open_check[dst_node->id_] = dst;
        }
      }
    }
    for (auto *p : pool) delete p;
    out_path.clear();
    for (auto &p : path) {
      out_path.push_back(p[0]);
      out_path.push_back(p[1]);
      out_path.push_back(p[2]);
    }
    return found;
  }

  // trg.cpp:692-730 (point_between == 1)
  static void refinePath(const std::vector<float> &in, std::vector<float> &out) {
    out.clear();
    size_t np = in.size() / 3;
    if (np == 0) return;
    std::deque<std::array<float, 3>> dense;
    for (size_t i = 0; i + 1 < np; ++i) {
      dense.push_back({in[3 * i], in[3 * i + 1], in[3 * i + 2]});
      dense.push_back({in[3 * i + 3], in[3 * i + 4], in[3 * i + 5]});
    }
    for (size_t i = 0; i < dense.size(); ++i) {
      if (i == dense.size() - 1) {
        out.insert(out.end(), dense[i].begin(), dense[i].end());
        break;
      }
      float sum[3] = {0.0, 0.0, 0.0};
      int cnt = 0;
      for (long j = (long)i - 1; j < (long)i + 2; ++j) {
        if (j < 0 || j >= (long)dense.size()) continue;
        for (int k = 0; k < 3; ++k) sum[k] += dense[j][k];
        cnt++;
      }
      for (int k = 0; k < 3; ++k) out.push_back(sum[k] / cnt);
    }
  }

  // ---- instrumentation -----------------------------------------------------
  struct Snapshot {
    std::vector<float> xyz;
    std::vector<int> state;
    std::vector<int> rowptr, col;
    std::vector<float> w, dist;
    std::vector<int> cid;  // creation id of node i of this snapshot
  };

  // graph in creation order before cleanGraph (ids == creation ids during initGraph)
  void snapshotPreClean() {
    pre_ = Snapshot();
    size_t V = all_created_.size();
    pre_.rowptr.push_back(0);
    for (size_t i = 0; i < V; ++i) {
      ONode *n = all_created_[i];
      pre_.xyz.insert(pre_.xyz.end(), n->pos_, n->pos_ + 3);
      pre_.state.push_back(n->state_);
      pre_.cid.push_back(n->cid_);
      for (OEdge *e : n->edges_) {
        pre_.col.push_back(e->dst_id_);
        pre_.w.push_back(e->weight_);
        pre_.dist.push_back(e->dist_);
      }
      pre_.rowptr.push_back((int)pre_.col.size());
    }
  }

  // current global graph, rows ordered by node id (ids are dense 0..V-1 after cleanGraph)
  Snapshot snapshotGlobal() {
    Snapshot s;
    OGraph &g = global_;
    int V = (int)g.nodes.size();
    s.rowptr.push_back(0);
    for (int id = 0; id < V; ++id) {
      auto it = g.nodes.find(id);
      if (it == g.nodes.end()) break;
      ONode *n = it->second;
      s.xyz.insert(s.xyz.end(), n->pos_, n->pos_ + 3);
      s.state.push_back(n->state_);
      s.cid.push_back(n->cid_);
      for (OEdge *e : n->edges_) {
        s.col.push_back(e->dst_id_);
        s.w.push_back(e->weight_);
        s.dist.push_back(e->dist_);
      }
      s.rowptr.push_back((int)s.col.size());
    }
    return s;
  }

  // Tiled-build extension (not in the reference): nodes may only be
  // created inside the core region, a sample outside it counts as a rejected (colliding) draw.
  // The default core is the whole plane, which leaves the reference behaviour untouched.
  float core_[4] = {-std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(),
                    std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity()};
  uint32_t epoch_base_ = 0;
  bool inCore(float x, float y) const {
    return x >= core_[0] && x < core_[2] && y >= core_[1] && y < core_[3];
  }

  OParams param_;
  KdApi kd_;  // backend captured at construction (own restatement or reference kdtree.c)
  OSampler sampler_;
  OCounters cnt_;
  OGraph global_, local_;
  uint32_t epoch_ = 0;
  int next_cid_ = 0;
  std::vector<ONode *> all_created_;
  Snapshot pre_, cur_;
  bool cov_f64_ = false;
  bool trace_wires_ = false;
  std::vector<WireTrace> wire_trace_;
  ONode *goal_node_ = nullptr;
  bool goal_known_ = false;
  float goal_pose2d_[2] = {0, 0};
};

}  // namespace

// ---------------------------------------------------------------------------
// C ABI for ctypes (tests / bench cpu_baseline only)
// ---------------------------------------------------------------------------
extern "C" {

// 0 = own restatement (okd.c); 1 = reference kdtree.c from `path` (oracle/_ref/libkdtree_ref.so).
// Must be called before any oracle is created.  Returns 0 on success.
int trg_oracle_set_kd_backend(int backend, const char *path) {
  if (backend == 0) {
    g_kd = make_own_api();
    g_kd_backend = 0;
    return 0;
  }
  void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return -1;
  KdApi a;
  g_ref_create = (ref_create_fn)dlsym(h, "kd_create");
  a.create = ref_create2;
  a.clear = (void (*)(void *))dlsym(h, "kd_clear");
  a.free_ = (void (*)(void *))dlsym(h, "kd_free");
  a.insert2 = (int (*)(void *, float, float, void *))dlsym(h, "kd_insert2");
  a.nearest2 = (void *(*)(void *, float, float))dlsym(h, "kd_nearest2");
  a.nearest_range2 = (void *(*)(void *, float, float, float))dlsym(h, "kd_nearest_range2");
  a.res_free = (void (*)(void *))dlsym(h, "kd_res_free");
  a.res_size = (int (*)(void *))dlsym(h, "kd_res_size");
  a.res_end = (int (*)(void *))dlsym(h, "kd_res_end");
  a.res_next = (int (*)(void *))dlsym(h, "kd_res_next");
  a.res_item_data = (void *(*)(void *))dlsym(h, "kd_res_item_data");
  if (!g_ref_create || !a.clear || !a.free_ || !a.insert2 || !a.nearest2 || !a.nearest_range2 ||
      !a.res_free || !a.res_size || !a.res_end || !a.res_next || !a.res_item_data) {
    dlclose(h);
    return -2;
  }
  g_ref_handle = h;
  g_kd = a;
  g_kd_backend = 1;
  return 0;
}
int trg_oracle_kd_backend() { return g_kd_backend; }

void *trg_oracle_create(float expand_dist, float robot_size, int sample_num, float height_threshold,
                        float collision_threshold, float update_collision_threshold,
                        float safety_factor, float goal_tolerance) {
  OParams p{expand_dist,         robot_size,
            sample_num,          height_threshold,
            collision_threshold, update_collision_threshold,
            safety_factor,       goal_tolerance};
  Oracle *o = new Oracle(p);
  o->sampler_.configure(0, 1, 16);
  return o;
}
void trg_oracle_destroy(void *h) { delete (Oracle *)h; }

void trg_oracle_set_sampler(void *h, int mode, uint32_t seed, int table_bits) {
  ((Oracle *)h)->sampler_.configure(mode, seed, table_bits);
}
void trg_oracle_set_tile(void *h, const float *core_xyxy, uint32_t epoch) {
  Oracle *o = (Oracle *)h;
  for (int i = 0; i < 4; ++i) o->core_[i] = core_xyxy[i];
  o->epoch_base_ = epoch;
}
void trg_oracle_set_cov_f64(void *h, int on) { ((Oracle *)h)->cov_f64_ = on != 0; }
void trg_oracle_set_trace(void *h, int on) { ((Oracle *)h)->trace_wires_ = on != 0; }

// direction table as the oracle's libm produced it (for cross-checking the engine's table)
void trg_oracle_get_table(void *h, float *cos_out, float *sin_out) {
  Oracle *o = (Oracle *)h;
  memcpy(cos_out, o->sampler_.cos_t.data(), o->sampler_.cos_t.size() * sizeof(float));
  memcpy(sin_out, o->sampler_.sin_t.data(), o->sampler_.sin_t.size() * sizeof(float));
}

void trg_oracle_set_global_map(void *h, const float *xyz, size_t n, size_t stride) {
  ((Oracle *)h)->setGlobalMap(xyz, n, stride);
}
void trg_oracle_set_local_map(void *h, float sx, float sy, const float *xyz, size_t n,
                              size_t stride) {
  ((Oracle *)h)->setLocalMap(sx, sy, xyz, n, stride);
}
int trg_oracle_init_graph(void *h, const float *start3d) {
  return ((Oracle *)h)->initGraph(start3d) ? 0 : -1;
}
void trg_oracle_update_graph(void *h) { ((Oracle *)h)->updateGraph(); }

// batched probes --------------------------------------------------------------
void trg_oracle_is_collision(void *h, int type, float threshold, const float *xy, size_t m,
                             int *flag, int *cnt, int *n) {
  Oracle *o = (Oracle *)h;
  for (size_t i = 0; i < m; ++i) {
    int c = 0, t = 0;
    flag[i] = o->isCollision(xy[2 * i], xy[2 * i + 1], type, threshold, &c, &t) ? 1 : 0;
    cnt[i] = c;
    n[i] = t;
  }
}
void trg_oracle_nearest_z(void *h, int type, const float *xy, size_t m, float *z) {
  Oracle *o = (Oracle *)h;
  for (size_t i = 0; i < m; ++i) z[i] = o->nearestZ(xy[2 * i], xy[2 * i + 1], type);
}
// pcl::VoxelGrid as TRGPlanner::loadPrebuiltMap uses it (trg_planner.cpp:91-94).  PCL is a third-party
// dependency the reference does not vendor or pin (find_package(PCL), cpp/trg_planner/CMakeLists.txt);
// this restates the published algorithm of pcl/filters/impl/voxel_grid.hpp (applyFilter, PCL 1.10-
// 1.14, downsample_all_data irrelevant for PointXYZ, min_points_per_voxel = 0): PARITY UNPINNED --
// the reference holds no fixture for it.  PCL's std::sort leaves the order inside a voxel
// unspecified; a stable sort (ascending point index) is used here.
// out: room for 3*n floats; returns the number of voxels, or n with *passthrough = 1 when the index
// space would overflow int32 (PCL warns and copies the input).
size_t trg_oracle_voxel_grid(const float *xyz, size_t n, float leaf, float *out, int *passthrough) {
  *passthrough = 0;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  size_t finite = 0;
  for (size_t i = 0; i < n; ++i) {
    const float *p = xyz + 3 * i;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    for (int a = 0; a < 3; ++a) {
      mn[a] = std::min(mn[a], p[a]);
      mx[a] = std::max(mx[a], p[a]);
    }
    finite++;
  }
  if (!finite) return 0;
  const float inv = 1.0f / leaf;
  const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1;
  const int64_t dy = (int64_t)((mx[1] - mn[1]) * inv) + 1;
  const int64_t dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) {
    memcpy(out, xyz, n * 3 * sizeof(float));
    *passthrough = 1;
    return n;
  }
  int min_b[3], max_b[3];
  for (int a = 0; a < 3; ++a) {
    min_b[a] = (int)std::floor(mn[a] * inv);
    max_b[a] = (int)std::floor(mx[a] * inv);
  }
  const int mul1 = max_b[0] - min_b[0] + 1;
  const int mul2 = mul1 * (max_b[1] - min_b[1] + 1);
  std::vector<std::pair<unsigned, unsigned>> iv;  // (voxel index, point index)
  iv.reserve(finite);
  for (size_t i = 0; i < n; ++i) {
    const float *p = xyz + 3 * i;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    const int i0 = (int)(std::floor(p[0] * inv) - (float)min_b[0]);
    const int i1 = (int)(std::floor(p[1] * inv) - (float)min_b[1]);
    const int i2 = (int)(std::floor(p[2] * inv) - (float)min_b[2]);
    iv.emplace_back((unsigned)(i0 + i1 * mul1 + i2 * mul2), (unsigned)i);
  }
  std::stable_sort(iv.begin(), iv.end(),
                   [](const std::pair<unsigned, unsigned> &a, const std::pair<unsigned, unsigned> &b) {
                     return a.first < b.first;
                   });
  size_t nv = 0;
  for (size_t s = 0; s < iv.size();) {
    size_t e = s;
    float cx = 0.0f, cy = 0.0f, cz = 0.0f;
    while (e < iv.size() && iv[e].first == iv[s].first) {
      const float *p = xyz + 3 * (size_t)iv[e].second;
      cx += p[0];
      cy += p[1];
      cz += p[2];
      ++e;
    }
    const float c = (float)(e - s);
    out[3 * nv] = cx / c;
    out[3 * nv + 1] = cy / c;
    out[3 * nv + 2] = cz / c;
    ++nv;
    s = e;
  }
  return nv;
}
void trg_oracle_is_frontier(void *h, const float *xy, size_t m, int *flag) {
  Oracle *o = (Oracle *)h;
  for (size_t i = 0; i < m; ++i) flag[i] = o->isFrontier(xy[2 * i], xy[2 * i + 1]) ? 1 : 0;
}
// p1,p2: m x 3 ; out: status, n_pts, weight, dist
void trg_oracle_edge_risk(void *h, int type, const float *p1, const float *p2, size_t m,
                          int *status, int *n_pts, float *weight, float *dist) {
  Oracle *o = (Oracle *)h;
  for (size_t i = 0; i < m; ++i) {
    status[i] = o->edgeRisk(p1 + 3 * i, p2 + 3 * i, type, weight[i], dist[i], n_pts[i]);
  }
}

// graph export ------------------------------------------------------------------
// which: 0 = current global graph (ids after cleanGraph), 1 = snapshot before the last cleanGraph
void trg_oracle_graph_sizes(void *h, int which, int *V, int *E) {
  Oracle *o = (Oracle *)h;
  if (which == 0) o->cur_ = o->snapshotGlobal();
  Oracle::Snapshot &s = which == 0 ? o->cur_ : o->pre_;
  *V = (int)s.state.size();
  *E = (int)s.col.size();
}
void trg_oracle_graph_export(void *h, int which, float *xyz, int *state, int *rowptr, int *col,
                             float *w, float *dist, int *cid) {
  Oracle *o = (Oracle *)h;
  Oracle::Snapshot &s = which == 0 ? o->cur_ : o->pre_;
  memcpy(xyz, s.xyz.data(), s.xyz.size() * sizeof(float));
  memcpy(state, s.state.data(), s.state.size() * sizeof(int));
  memcpy(rowptr, s.rowptr.data(), s.rowptr.size() * sizeof(int));
  memcpy(col, s.col.data(), s.col.size() * sizeof(int));
  memcpy(w, s.w.data(), s.w.size() * sizeof(float));
  memcpy(dist, s.dist.data(), s.dist.size() * sizeof(float));
  if (cid) memcpy(cid, s.cid.data(), s.cid.size() * sizeof(int));
}

size_t trg_oracle_trace_size(void *h) { return ((Oracle *)h)->wire_trace_.size(); }
void trg_oracle_trace_export(void *h, int *src, int *dst, int *status, int *n_pts, float *weight,
                             float *dist) {
  Oracle *o = (Oracle *)h;
  for (size_t i = 0; i < o->wire_trace_.size(); ++i) {
    const WireTrace &t = o->wire_trace_[i];
    src[i] = t.src_cid;
    dst[i] = t.dst_cid;
    status[i] = t.status;
    n_pts[i] = t.n_pts;
    weight[i] = t.weight;
    dist[i] = t.dist;
  }
}

// counters: 20 x uint64 in the order of OCounters
void trg_oracle_counters(void *h, uint64_t *out) {
  Oracle *o = (Oracle *)h;
  const OCounters &c = o->cnt_;
  uint64_t v[] = {c.collision_queries, c.collision_hits, c.nn_map_queries, c.ellipse_queries,
                  c.ellipse_hits,      c.wire_calls,     c.wire_evals,     c.wire_ok,
                  c.wire_gate,         c.wire_seg,       c.wire_empty,     c.wire_few,
                  c.wire_clamped,      c.expanded,       c.trials,         c.samples,
                  c.created,           c.invalid_created, c.nn_node_queries, c.sample_hits,
                  c.wire_hits_new,     c.wire_hits_other};
  memcpy(out, v, sizeof(v));
}
void trg_oracle_reset_counters(void *h) { ((Oracle *)h)->cnt_ = OCounters(); }

// planning ------------------------------------------------------------------------
// returns number of path points (0 = not found); info = {direct_dist, path_length, avg_risk}
int trg_oracle_plan(void *h, const float *start2d, const float *goal3d, float *path_xyz,
                    int max_pts, float *info) {
  Oracle *o = (Oracle *)h;
  std::vector<float> path;
  float dd = 0, pl = 0, ar = 0;
  bool ok = o->planSafePath(start2d, goal3d, path, dd, pl, ar);
  info[0] = dd;
  info[1] = pl;
  info[2] = ar;
  if (!ok) return 0;
  int np = (int)(path.size() / 3);
  if (np > max_pts) np = max_pts;
  memcpy(path_xyz, path.data(), (size_t)np * 3 * sizeof(float));
  return (int)(path.size() / 3);
}
int trg_oracle_refine(const float *in_xyz, int np, float *out_xyz, int max_pts) {
  std::vector<float> in(in_xyz, in_xyz + 3 * (size_t)np), out;
  Oracle::refinePath(in, out);
  int no = (int)(out.size() / 3);
  memcpy(out_xyz, out.data(), (size_t)std::min(no, max_pts) * 3 * sizeof(float));
  return no;
}

// 3x3 SVD probe (fp32): A row-major in, U row-major out
void trg_oracle_svd_u3(const float *A, float *U) {
  float a[3][3], u[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) a[r][c] = A[3 * r + c];
  jacobi_svd_u3(a, u);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) U[3 * r + c] = u[r][c];
}

}  // extern "C"
