// CPU check: trg::MapOrderSim against the real std::unordered_map<int,int> for the operation
// sequences the engine performs (fill dense keys, iterate, build a second map in iteration order,
// copy-assign it, clear, refill with a different size, ...), including bucket counts.
#include <cstdio>
#include <unordered_map>
#include <vector>

#include "../../trg-planner_amd/csrc/map_order_sim.h"

static bool same_order(const std::unordered_map<int, int> &m, const trg::MapOrderSim &s) {
  std::vector<int> want, got;
  for (auto &kv : m) want.push_back(kv.first);
  s.iteration_order(got);
  return want == got && m.bucket_count() == s.bucket_count();
}

int main() {
  const int sizes[] = {1, 2, 11, 12, 13, 14, 29, 30, 59, 60, 1000, 5, 70000, 3, 123457, 123456, 40};
  std::unordered_map<int, int> nodes;  // trgStruct::nodes
  trg::MapOrderSim sim;
  unsigned rng = 12345u;
  for (int round = 0; round < 3; ++round) {
    for (int V : sizes) {
      // resetGraph: clear(); addNode: nodes[id] = node
      nodes.clear();
      sim.clear();
      for (int i = 0; i < V; ++i) {
        nodes[i] = i;
        sim.insert_next();
        if ((i < 64 || i % 997 == 0) && !same_order(nodes, sim)) {
          printf("fill mismatch V=%d i=%d\n", V, i);
          return 1;
        }
      }
      if (!same_order(nodes, sim)) {
        printf("fill mismatch V=%d\n", V);
        return 1;
      }
      // cleanGraph: new_nodes[new_id] for the kept subset, in iteration order; nodes = new_nodes
      std::unordered_map<int, int> new_nodes;
      trg::MapOrderSim new_sim;
      int new_id = 0;
      for (auto &kv : nodes) {
        rng = rng * 1664525u + 1013904223u;
        if ((rng >> 28) == 0 && kv.first != 0) continue;  // drop ~6%
        new_nodes[new_id] = kv.first;
        new_sim.insert_next();
        new_id++;
      }
      if (!same_order(new_nodes, new_sim)) {
        printf("new_nodes mismatch V=%d\n", V);
        return 1;
      }
      nodes = new_nodes;
      sim.assign_from(new_sim);
      if (!same_order(nodes, sim)) {
        printf("assign mismatch V=%d\n", V);
        return 1;
      }
      // adopting the bucket state of the real map must reproduce later behaviour too
      trg::MapOrderSim adopted;
      adopted.adopt_bucket_state(nodes);
      std::unordered_map<int, int> copy = nodes;
      copy.clear();
      adopted.clear();
      for (int i = 0; i < V / 2 + 3; ++i) {
        copy[i] = i;
        adopted.insert_next();
      }
      if (!same_order(copy, adopted)) {
        printf("adopt mismatch V=%d\n", V);
        return 1;
      }
    }
  }
  // updateGraph cycles (trg.cpp:456-489): after the copy assignment the graph grows by appended
  // keys (nodes[node_id] = node for the next dense ids), then cleanGraph renumbers again -- several
  // cycles on the same container, appended counts crossing rehash thresholds
  {
    std::unordered_map<int, int> nodes2;
    trg::MapOrderSim sim2;
    for (int i = 0; i < 64736; ++i) {
      nodes2[i] = i;
      sim2.insert_next();
    }
    const int grow[] = {0, 1, 7, 300, 2500, 40000, 3, 90000, 12};
    for (int cyc = 0; cyc < 9; ++cyc) {
      // cleanGraph
      std::unordered_map<int, int> new_nodes;
      trg::MapOrderSim new_sim;
      int new_id = 0;
      for (auto &kv : nodes2) {
        rng = rng * 1664525u + 1013904223u;
        if ((rng >> 27) == 0 && kv.first != 0) continue;  // drop ~3%
        new_nodes[new_id] = kv.first;
        new_sim.insert_next();
        new_id++;
      }
      nodes2 = new_nodes;
      sim2.assign_from(new_sim);
      if (!same_order(nodes2, sim2)) {
        printf("update cycle %d: assign mismatch\n", cyc);
        return 1;
      }
      // the update appends nodes
      for (int k = 0; k < grow[cyc]; ++k) {
        const int id = (int)nodes2.size();
        nodes2[id] = id;
        sim2.insert_next();
        if ((k < 32 || k % 4099 == 0) && !same_order(nodes2, sim2)) {
          printf("update cycle %d: append mismatch at %d\n", cyc, k);
          return 1;
        }
      }
      if (!same_order(nodes2, sim2)) {
        printf("update cycle %d: append mismatch\n", cyc);
        return 1;
      }
    }
  }
  printf("ok\n");
  return 0;
}
