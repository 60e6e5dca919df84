// Host-logic check (CPU): the engine's node indices (trg-planner_amd/csrc/host_index.h) against
// the oracle's kd-tree restatement (oracle/okd.c) on the same insertion sequence.
//   NodeKd   : identical 1-NN winners and identical range-hit ORDER (tree-shape dependent)
//   NodeGrid : identical 1-NN winner whenever it reports no tie
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../oracle/okd.h"
#include "../../trg-planner_amd/csrc/host_index.h"

int main() {
  std::mt19937 gen(12345);
  std::uniform_real_distribution<float> U(0.0f, 60.0f);
  trg::NodeKd kd;
  trg::NodeGrid grid;
  grid.reset(0, 0, 60, 60, 0.3f);
  okdtree *ref = okd_create();
  std::vector<float> xs, ys;
  long ties = 0, checked = 0, multi = 0;
  for (int i = 0; i < 20000; ++i) {
    float x = U(gen), y = U(gen);
    if (i % 9 == 0) {  // lattice points: exact ties
      x = 0.5f * (float)(int)(x * 2);
      y = 0.5f * (float)(int)(y * 2);
    }
    xs.push_back(x);
    ys.push_back(y);
    kd.insert(x, y, i);
    grid.insert(x, y);
    okd_insert2(ref, x, y, (void *)(size_t)(i + 1));
    if (i % 3) continue;
    float qx = U(gen), qy = U(gen);
    if (i % 12 == 0) {
      qx = 0.25f * (float)(int)(qx * 4);
      qy = 0.25f * (float)(int)(qy * 4);
    }
    okdres *r = okd_nearest2(ref, qx, qy);
    const int want = (int)(size_t)okd_res_item_data(r) - 1;
    okd_res_free(r);
    if (kd.nearest(qx, qy) != want) {
      printf("NodeKd nearest mismatch at %d\n", i);
      return 1;
    }
    bool tie = false;
    const int g = grid.nearest(qx, qy, &tie);
    if (tie) {
      ++ties;
      // the tree-free exact tie-break must name the node the kd-tree returns
      std::vector<int> tied;
      grid.tied_set(qx, qy, grid.dist2(g, qx, qy), tied);
      const int w = trg::kd_tie_winner(xs.data(), ys.data(), (int)xs.size(), qx, qy, tied);
      if (tied.size() < 2 || w != want) {
        printf("tie-break mismatch at %d: %d vs %d (tied %zu)\n", i, w, want, tied.size());
        return 1;
      }
    } else if (g != want) {
      printf("NodeGrid nearest mismatch at %d: %d vs %d\n", i, g, want);
      return 1;
    }
    for (float rad : {0.3f, 0.4f, 1.1f}) {
      std::vector<int> got;
      kd.range(qx, qy, rad, got);
      okdres *rr = okd_nearest_range2(ref, qx, qy, rad);
      size_t k = 0;
      while (!okd_res_end(rr)) {
        const int id = (int)(size_t)okd_res_item_data(rr) - 1;
        if (k >= got.size() || got[k] != id) {
          printf("NodeKd range order mismatch at %d\n", i);
          return 1;
        }
        ++k;
        okd_res_next(rr);
      }
      if (k != got.size()) {
        printf("NodeKd range size mismatch at %d\n", i);
        return 1;
      }
      okd_res_free(rr);
      // the tree-free way the planner finds the reference's FIRST hit (= reached last by the walk): the
      // hit set from the grid, the order from kd_range_first_of_two
      std::vector<int> set;
      bool doubt = false;
      grid.range_set(qx, qy, rad, set, &doubt);
      if (!doubt) {
        if (set.size() != got.size()) {
          printf("NodeGrid range_set size mismatch at %d: %zu vs %zu\n", i, set.size(), got.size());
          return 1;
        }
        if (set.size() >= 2) {
          int last = set[0];
          for (size_t t = 1; t < set.size(); ++t)
            if (trg::kd_range_first_of_two(xs.data(), ys.data(), (int)xs.size(), qx, qy, last, set[t]) == last)
              last = set[t];
          if (last != got[0]) {
            printf("range first-hit mismatch at %d: %d vs %d (hits %zu)\n", i, last, got[0], set.size());
            return 1;
          }
          ++multi;
        }
      }
    }
    ++checked;
  }
  okd_free(ref);
  printf("ok checked=%ld ties=%ld multi_hit_orders=%ld\n", checked, ties, multi);
  return (ties > 0 && multi > 100) ? 0 : 2;  // the data must actually contain ties and multi-hit queries
}
