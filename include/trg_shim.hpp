// include/trg_shim.hpp -- C++ `TRG` class over the C ABI of include/trg_engine.h.
//
// Same class name, nested types, method names, argument order and meaning as the reference's
// `class TRG` (cpp/trg_planner/core/trg_planner/include/graph/trg.h:18-98), so that code written
// against it (TRGPlanner, the ROS nodes, the pybind module) keeps compiling with the engine
// underneath.  Header-only; link with libtrg_engine.so.
//
// Eigen and PCL are not part of this repository's toolchain, so the members are Eigen-free:
//   Eigen::Vector2f / Vector3f  -> trg_amd::Vec2f / Vec3f  (std::array<float, N>, same memory layout)
//   PointCloudPtr&              -> (const float *xyz, size_t n, size_t stride); for
//                                  pcl::PointCloud<pcl::PointXYZ> that is
//                                  (reinterpret_cast<const float *>(map->points.data()), map->size(), 4)
// INTEGRATION.md shows the three-line adapters a reference maintainer adds for the Eigen/PCL
// signatures.  This file is compiled in this repository: trg-planner_amd/csrc/trg_pybind.cpp binds
// it (module trg_planner._trg_pybind) and tests/cpp/shim_check.cpp instantiates every member.
//
// Members the engine has no use for are kept as documented no-ops where a caller of the reference
// could reach them (addNode / wireEdge / expandGraph / cleanGraph / setLocalGraph are steps of
// initGraph / updateGraph inside the engine).
#ifndef TRG_SHIM_HPP_
#define TRG_SHIM_HPP_

#include <array>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "trg_engine.h"

namespace trg_amd {

using Vec2f = std::array<float, 2>;
using Vec3f = std::array<float, 3>;

class TRG {
 public:
  struct Edge {  // trg.h:20-25
    Edge(int dst_id, float weight, float dist) : dst_id_(dst_id), weight_(weight), dist_(dist) {}
    int dst_id_;
    float weight_;
    float dist_;
  };

  enum struct NodeState {  // trg.h:27-31
    Valid = 0,
    Invalid = -1,
    Frontier = 1,
  };

  struct Node {  // trg.h:33-40 (edges are shared so that copies handed to Python alias them)
    Node(int id, const Vec2f &pos2d, float z, NodeState state)
        : id_(id), pos_{pos2d[0], pos2d[1], z}, state_(state) {}
    int id_;
    Vec3f pos_;
    NodeState state_;
    std::vector<std::shared_ptr<Edge>> edges_;
  };
  using NodeMap = std::unordered_map<int, std::shared_ptr<Node>>;

  // trg.h:51-59 / trg.cpp:11-34; device = HIP device ordinal (an addition with a default)
  TRG(bool isVerbose, float expand_dist, float robot_size, int sample_num, float height_threshold,
      float collision_threshold, float update_collision_threshold, float safety_factor,
      float goal_tolerance, int device = 0) {
    TrgParams p;
    p.is_verbose = isVerbose ? 1 : 0;
    p.expand_dist = expand_dist;
    p.robot_size = robot_size;
    p.sample_num = sample_num;
    p.height_threshold = height_threshold;
    p.collision_threshold = collision_threshold;
    p.update_collision_threshold = update_collision_threshold;
    p.safety_factor = safety_factor;
    p.goal_tolerance = goal_tolerance;
    const TrgStatus st = trg_engine_create(&p, device, &e_);
    if (st != TRG_OK) {
      std::string msg = e_ ? trg_engine_last_error(e_) : "trg_engine_create failed";
      if (e_) trg_engine_destroy(e_);
      e_ = nullptr;
      throw std::runtime_error(status_name(st) + ": " + msg);
    }
    sampler_.seed = 1;
    sampler_.table_bits = 16;
  }
  virtual ~TRG() {
    if (e_) trg_engine_destroy(e_);
  }
  TRG(const TRG &) = delete;
  TRG &operator=(const TRG &) = delete;

  // The reference seeds std::mt19937 from std::random_device (trg.cpp:20); here the sample stream is
  // an explicit input (include/trg_engine.h, TrgSampler).
  void setSampler(uint32_t seed, int table_bits = 16) {
    sampler_.seed = seed;
    sampler_.table_bits = table_bits;
  }

  void initGraph(bool /*isPreMap*/, Vec3f start3d) {  // trg.cpp:36-64
    std::lock_guard<std::mutex> lock(mtx.graph);
    check(trg_engine_init_graph(e_, start3d.data(), &sampler_));
  }
  void loadPrebuiltGraph(const std::string &filepath) {  // trg.cpp:66-128
    std::lock_guard<std::mutex> lock(mtx.graph);
    check(trg_engine_load_json(e_, filepath.c_str()));
  }
  void saveGraph(const std::string &filepath) {  // trg.cpp:130-177
    std::lock_guard<std::mutex> lock(mtx.graph);
    check(trg_engine_save_json(e_, filepath.c_str()));
  }

  void setGlobalMap(const float *xyz, size_t n, size_t stride) {  // trg.cpp:179-193 (not locked there either)
    check(trg_engine_set_global_map(e_, xyz, n, stride));
  }
  void setLocalMap(Vec2f start2d, const float *xyz, size_t n, size_t stride) {  // trg.cpp:195-209
    std::lock_guard<std::mutex> lock(mtx.graph);
    check(trg_engine_set_local_map(e_, start2d.data(), xyz, n, stride));
  }
  void setLocalGraph(bool /*useMutex*/) {}  // trg.cpp:211-231: done by setLocalMap / updateGraph inside the engine

  void updateGraph() {  // trg.cpp:456-489
    std::lock_guard<std::mutex> lock(mtx.graph);
    check(trg_engine_update_graph(e_));
  }

  void setGoal(Vec3f &goal) { goal_ = goal; }  // trg.cpp:537-565: resolved by planSafePath inside the engine
  bool checkReadched(Vec2f &pos2d) {          // (sic) trg.cpp:567-574
    return trg_engine_check_reached(e_, pos2d.data()) != 0;
  }
  bool checkReplan(Vec2f &pos2d, std::vector<Vec3f> &path) {  // trg.cpp:576-601
    return trg_engine_check_replan(e_, pos2d.data(), path.empty() ? nullptr : path[0].data(),
                                   (int32_t)path.size()) != 0;
  }

  // trg.cpp:603-690
  bool planSafePath(Vec2f &start2d, Vec3f &goal_pose, std::vector<Vec3f> &out_path,
                    float &direct_dist, float &path_length, float &avg_risk) {
    std::lock_guard<std::mutex> lock(mtx.graph);
    goal_ = goal_pose;
    TrgPathInfo info{};
    std::vector<Vec3f> buf(4096);
    TrgStatus st = trg_engine_plan(e_, start2d.data(), goal_pose.data(), buf[0].data(),
                                   (int32_t)buf.size(), &info);
    if (st == TRG_OK && info.num_points > (int32_t)buf.size()) {
      buf.resize((size_t)info.num_points);
      st = trg_engine_plan(e_, start2d.data(), goal_pose.data(), buf[0].data(), (int32_t)buf.size(),
                           &info);
    }
    if (st == TRG_ERR_NOT_FOUND) return false;
    check(st);
    buf.resize((size_t)info.num_points);
    out_path = std::move(buf);
    direct_dist = info.direct_dist;
    path_length = info.path_length;
    avg_risk = info.avg_risk;
    return true;
  }
  void refinePath(std::vector<Vec3f> &in_path, std::vector<Vec3f> &out_path) {  // trg.cpp:692-730
    out_path.assign(2 * in_path.size() + 2, Vec3f{0, 0, 0});
    const int32_t n = trg_engine_refine_path(in_path.empty() ? nullptr : in_path[0].data(),
                                             (int32_t)in_path.size(), out_path[0].data(),
                                             (int32_t)out_path.size());
    out_path.resize((size_t)(n > 0 ? n : 0));
  }

  void resetGraph(std::string type) { check(trg_engine_reset_graph(e_, kind(type))); }  // trg.cpp:732-737
  void resetMap(std::string type) { check(trg_engine_reset_map(e_, kind(type))); }      // trg.cpp:739-744

  bool isCollision(Vec2f &pos2d, std::string type, float threshold) {  // trg.cpp:746-778
    int32_t flag = 0;
    check(trg_engine_is_collision_batch(e_, kind(type), threshold, pos2d.data(), 1, &flag, nullptr,
                                        nullptr));
    return flag != 0;
  }
  bool isFrontier(Vec2f &pos2d) {  // trg.cpp:780-803
    int32_t flag = 0;
    check(trg_engine_is_frontier_batch(e_, pos2d.data(), 1, &flag));
    return flag != 0;
  }

  // trg.cpp:805-824.  The engine's graph lives in CSR arrays; both accessors build node objects
  // from it (ids as the reference numbers them; local graph: ids of the global nodes).
  NodeMap getGraph(std::string type) { return build_nodes(type); }
  NodeMap getGraphCopy(std::string type) {
    std::lock_guard<std::mutex> lock(mtx.graph);
    return build_nodes(type);
  }
  void lockGraph() { mtx.graph.lock(); }      // trg.cpp:826
  void unlockGraph() { mtx.graph.unlock(); }  // trg.cpp:827-828

  // the CSR view itself, for callers that do not need node objects
  TrgCsrView getGraphCSR(const std::string &type) {
    TrgCsrView v{};
    check(trg_engine_export_csr(e_, kind(type), &v));
    return v;
  }
  TrgEngine *engine() { return e_; }

 protected:
  static std::string status_name(TrgStatus st) {
    static const char *names[] = {"TRG_OK",          "TRG_ERR_INVALID_ARG", "TRG_ERR_NO_MAP",
                                  "TRG_ERR_NO_ROOT", "TRG_ERR_DEVICE",      "TRG_ERR_NO_GRAPH",
                                  "TRG_ERR_NOT_FOUND", "TRG_ERR_IO",        "TRG_ERR_CAPACITY"};
    return ((int)st >= 0 && (int)st <= 8) ? names[(int)st] : "TRG_ERR_?";
  }
  void check(TrgStatus st) const {
    if (st != TRG_OK) throw std::runtime_error(status_name(st) + ": " + trg_engine_last_error(e_));
  }
  static TrgKind kind(const std::string &type) {
    if (type == "global") return TRG_KIND_GLOBAL;
    if (type == "local") return TRG_KIND_LOCAL;
    if (type == "preclean") return TRG_KIND_PRECLEAN;
    throw std::runtime_error("TRG_ERR_INVALID_ARG: unknown graph type " + type);
  }
  NodeMap build_nodes(const std::string &type) {
    TrgCsrView v{};
    check(trg_engine_export_csr(e_, kind(type), &v));
    NodeMap out;
    out.reserve((size_t)v.num_nodes);
    const bool local = type == "local";
    for (int32_t row = 0; row < v.num_nodes; ++row) {
      const int id = local ? v.creation_id[row] : row;
      Vec2f p2{v.node_xyz[3 * row], v.node_xyz[3 * row + 1]};
      auto n = std::make_shared<Node>(id, p2, v.node_xyz[3 * row + 2], (NodeState)v.node_state[row]);
      n->edges_.reserve((size_t)(v.rowptr[row + 1] - v.rowptr[row]));
      for (int32_t k = v.rowptr[row]; k < v.rowptr[row + 1]; ++k)
        n->edges_.push_back(std::make_shared<Edge>(v.col[k], v.weight[k], v.dist[k]));
      out.emplace(id, std::move(n));
    }
    return out;
  }

  TrgEngine *e_ = nullptr;
  TrgSampler sampler_{};
  Vec3f goal_{0, 0, 0};
  struct Mutex {  // trg.h:143-145
    std::mutex graph;
  } mtx;
};

}  // namespace trg_amd

#endif  // TRG_SHIM_HPP_
