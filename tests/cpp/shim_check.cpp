// CPU check of include/trg_shim.hpp: every member of the C++ `TRG` shim is instantiated (so the
// header really compiles and links against libtrg_engine.so), and without a GPU the constructor
// reports TRG_ERR_DEVICE instead of falling back to anything.  With a GPU (argv[1] == "gpu") a tiny
// map is built and the accessors are exercised end to end.
#include <cstdio>
#include <cstring>
#include <random>
#include <string>

#include "../../include/trg_shim.hpp"

using trg_amd::TRG;
using trg_amd::Vec2f;
using trg_amd::Vec3f;

// never called on the CPU path, but forces the compiler to instantiate every member
static int use_all(TRG &t, const float *xyz, size_t n) {
  t.setSampler(7, 16);
  t.setGlobalMap(xyz, n, 3);
  t.initGraph(false, Vec3f{5.0f, 5.0f, 0.0f});
  auto g = t.getGraphCopy("global");
  auto g2 = t.getGraph("global");
  if (g.size() != g2.size() || g.empty()) return 1;
  size_t deg = 0;
  for (auto &kv : g) {
    if (kv.second->id_ != kv.first || kv.second->state_ == TRG::NodeState::Invalid) return 2;
    deg += kv.second->edges_.size();
  }
  const TrgCsrView v = t.getGraphCSR("global");
  if ((size_t)v.num_nodes != g.size() || (size_t)v.num_edges != deg) return 3;
  Vec2f s{5.0f, 5.0f};
  Vec3f goal = g.begin()->second->pos_;
  std::vector<Vec3f> path, smooth;
  float direct = 0, len = 0, risk = 0;
  t.setGoal(goal);
  const bool found = t.planSafePath(s, goal, path, direct, len, risk);
  if (found) t.refinePath(path, smooth);
  (void)t.checkReadched(s);
  (void)t.checkReplan(s, path);
  (void)t.isCollision(s, "global", 0.1f);
  t.setLocalMap(s, xyz, n, 3);
  t.setLocalGraph(false);
  (void)t.isFrontier(s);
  t.updateGraph();
  t.lockGraph();
  t.unlockGraph();
  t.saveGraph("/tmp/trg_shim_check_graph.json");
  t.loadPrebuiltGraph("/tmp/trg_shim_check_graph.json");
  if (t.getGraphCopy("global").empty()) return 4;
  t.resetGraph("local");
  t.resetMap("local");
  return 0;
}

int main(int argc, char **argv) {
  const bool want_gpu = argc > 1 && !strcmp(argv[1], "gpu");
  try {
    TRG t(false, 0.6f, 0.3f, 7, 0.16f, 0.1f, 0.5f, 3.0f, 0.8f);
    // flat 10 m x 10 m patch, 0.1 m spacing with jitter
    std::mt19937 gen(3);
    std::uniform_real_distribution<float> J(-0.02f, 0.02f);
    std::vector<float> xyz;
    for (int i = 0; i < 100; ++i)
      for (int j = 0; j < 100; ++j) {
        xyz.push_back(0.1f * i + J(gen));
        xyz.push_back(0.1f * j + J(gen));
        xyz.push_back(0.01f * J(gen));
      }
    const int rc = use_all(t, xyz.data(), xyz.size() / 3);
    if (rc) {
      printf("FAIL: use_all -> %d\n", rc);
      return 1;
    }
    printf("ok gpu\n");
    return 0;
  } catch (const std::exception &ex) {
    if (!want_gpu && strstr(ex.what(), "TRG_ERR_DEVICE")) {
      printf("ok no-gpu (%s)\n", ex.what());
      return 0;
    }
    printf("FAIL: %s\n", ex.what());
    return 1;
  }
}
