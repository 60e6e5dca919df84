// host_index.h -- host-side node indices used by the sequential BFS replay.
//
// The reference keeps graph nodes in a second kd-tree (trgStruct::node_tree, trg.h:106) and asks
// it three things: nearest node (trg.cpp:408, 615), nodes within a radius (trg.cpp:430, 544, 593,
// 788) and, implicitly through the result-list order, WHICH hit comes first.  Two structures
// answer those here:
//   NodeGrid  -- uniform hash grid, O(1) nearest-node lookups for the replay's hot loop.  It
//                returns the same node as the kd-tree whenever the fp32 minimum is unique and
//                reports ties so the caller can fall back.
//   NodeKd    -- index-based replica of the reference tree's shape (same insertion rule, same
//                traversal), used where the answer depends on tree shape: range-query ORDER
//                (kdtree.c:282, results are pushed at the list head) and nearest-neighbour ties
//                (kdtree.c:343, first strictly-nearer in traversal order).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace trg {

class NodeKd {
 public:
  void clear() {
    x_.clear();
    y_.clear();
    lo_.clear();
    hi_.clear();
    axis_.clear();
    payload_.clear();
    have_box_ = false;
  }
  size_t size() const { return x_.size(); }

  // kdtree.c:167-194 (+ bounding box :202-206)
  void insert(float x, float y, int payload) {
    const int me = (int)x_.size();
    x_.push_back(x);
    y_.push_back(y);
    lo_.push_back(-1);
    hi_.push_back(-1);
    payload_.push_back(payload);
    int axis = 0;
    if (me > 0) {
      int cur = 0;
      for (;;) {
        const int ax = axis_[cur];
        const bool low = ax == 0 ? (x < x_[cur]) : (y < y_[cur]);
        int &slot = low ? lo_[cur] : hi_[cur];
        axis = (ax + 1) % 2;
        if (slot < 0) {
          slot = me;
          break;
        }
        cur = slot;
      }
    }
    axis_.push_back((int8_t)axis);
    if (!have_box_) {
      bmin_[0] = bmax_[0] = x;
      bmin_[1] = bmax_[1] = y;
      have_box_ = true;
    } else {
      if (x < bmin_[0]) bmin_[0] = x;
      if (x > bmax_[0]) bmax_[0] = x;
      if (y < bmin_[1]) bmin_[1] = y;
      if (y > bmax_[1]) bmax_[1] = y;
    }
  }

  // kdtree.c:364-417; returns the payload of the winner, -1 if the tree is empty
  int nearest(float qx, float qy) const {
    if (x_.empty()) return -1;
    const float q[2] = {qx, qy};
    int best = 0;
    float best_d2 = d2(0, q);
    float box[4] = {bmin_[0], bmin_[1], bmax_[0], bmax_[1]};
    nn_walk(0, q, best, best_d2, box, box + 2);
    return payload_[best];
  }

  // kdtree.c:479-501; payloads in the order the reference's result iterator yields them
  void range(float qx, float qy, float r, std::vector<int> &out) const {
    out.clear();
    if (x_.empty()) return;
    const float q[2] = {qx, qy};
    range_walk(0, q, r, out);
    // discovery order was appended; the reference iterates newest-first
    for (size_t i = 0, j = out.size(); i + 1 < j; ++i) {
      --j;
      std::swap(out[i], out[j]);
    }
  }

 private:
  float d2(int n, const float *q) const {
    float acc = 0;
    acc += (x_[n] - q[0]) * (x_[n] - q[0]);
    acc += (y_[n] - q[1]) * (y_[n] - q[1]);
    return acc;
  }
  float coord(int n, int ax) const { return ax == 0 ? x_[n] : y_[n]; }

  static float box_d2(const float *bmin, const float *bmax, const float *q) {
    float acc = 0;
    for (int i = 0; i < 2; ++i) {
      if (q[i] < bmin[i]) {
        acc += (bmin[i] - q[i]) * (bmin[i] - q[i]);
      } else if (q[i] > bmax[i]) {
        acc += (bmax[i] - q[i]) * (bmax[i] - q[i]);
      }
    }
    return acc;
  }

  void nn_walk(int n, const float *q, int &best, float &best_d2, float *bmin, float *bmax) const {
    const int ax = axis_[n];
    int nearer, farther;
    float *near_edge, *far_edge;
    if (q[ax] - coord(n, ax) <= 0) {
      nearer = lo_[n];
      farther = hi_[n];
      near_edge = bmax + ax;
      far_edge = bmin + ax;
    } else {
      nearer = hi_[n];
      farther = lo_[n];
      near_edge = bmin + ax;
      far_edge = bmax + ax;
    }
    if (nearer >= 0) {
      const float keep = *near_edge;
      *near_edge = coord(n, ax);
      nn_walk(nearer, q, best, best_d2, bmin, bmax);
      *near_edge = keep;
    }
    const float dd = d2(n, q);
    if (dd < best_d2) {
      best = n;
      best_d2 = dd;
    }
    if (farther >= 0) {
      const float keep = *far_edge;
      *far_edge = coord(n, ax);
      if (box_d2(bmin, bmax, q) < best_d2) nn_walk(farther, q, best, best_d2, bmin, bmax);
      *far_edge = keep;
    }
  }

  void range_walk(int n, const float *q, float r, std::vector<int> &out) const {
    if (n < 0) return;
    if (d2(n, q) <= r * r) out.push_back(payload_[n]);
    const float dx = q[axis_[n]] - coord(n, axis_[n]);
    range_walk(dx <= 0.0 ? lo_[n] : hi_[n], q, r, out);
    if (std::fabs((double)dx) < r) range_walk(dx <= 0.0 ? hi_[n] : lo_[n], q, r, out);
  }

  std::vector<float> x_, y_;
  std::vector<int> lo_, hi_, payload_;
  std::vector<int8_t> axis_;
  bool have_box_ = false;
  float bmin_[2] = {0, 0}, bmax_[2] = {0, 0};
};

// ---- exact nearest-neighbour tie-break without the tree ---------------------------------------------
// kd_nearest keeps the FIRST node it visits among those at the minimal fp32 squared distance
// (strict `<` updates, kdtree.c:343; the root is the initial best, kdtree.c:393-396), and it visits
// nearer subtree -> node -> farther subtree (kdtree.c:327-361).  A pruned subtree never holds the
// first of the tied nodes (pruning needs an equally near node already found), so the winner is the
// root if it is among the tied, else the tied node that comes first in the unpruned traversal.
// Which of two nodes A, B comes first is decided at their lowest common ancestor in the insertion
// tree.  The tree is never built: a subtree's root is the node with the smallest insertion index
// inside the subtree's region, so the common path is followed by repeatedly taking the minimum
// index of a shrinking candidate list.  Cost O(sum of list sizes); used only on exact ties.
//   px, py: positions in insertion order; only indices < K exist at query time.
inline int kd_first_of_two(const float *px, const float *py, int K, float qx, float qy, int A, int B) {
  if (A == B) return A;
  const float q[2] = {qx, qy};
  auto coord = [&](int n, int ax) { return ax == 0 ? px[n] : py[n]; };
  // The common path is walked with ONE forward scan over the insertion order: the subtree below
  // `cur` on the side of A and B is the half-open box [lo, hi); its root is the first later node
  // inside the box (children are inserted after their parents).  O(K), no allocation.
  float lo[2] = {-INFINITY, -INFINITY}, hi[2] = {INFINITY, INFINITY};
  int cur = 0, axis = 0;
  for (;;) {
    const float split = coord(cur, axis);
    const bool near_is_left = (q[axis] - split) <= 0;
    if (cur == A || cur == B) {
      const int other = (cur == A) ? B : A;
      const bool other_left = coord(other, axis) < split;
      // other lies in cur's subtree: visited before cur iff it is in the nearer subtree
      return (other_left == near_is_left) ? other : cur;
    }
    const bool a_left = coord(A, axis) < split, b_left = coord(B, axis) < split;
    if (a_left != b_left) return (a_left == near_is_left) ? A : B;
    // same side: descend into it (kd_insert sends `<` to the left, kdtree.c:190)
    if (a_left)
      hi[axis] = split;
    else
      lo[axis] = split;
    axis ^= 1;
    int n = cur + 1;
    for (; n < K; ++n)
      if (px[n] >= lo[0] && px[n] < hi[0] && py[n] >= lo[1] && py[n] < hi[1]) break;
    if (n >= K) return A < B ? A : B;  // cannot happen: A and B are inside the box
    cur = n;
  }
}

// Which of two nodes, both inside the radius of a kd_nearest_range query at (qx, qy), does the walk reach
// first?  find_nearest (kdtree.c:270-301) tests the node itself, then the subtree on the query's side of
// the split, then (if |dx| < range) the other one: pre-order, so an ancestor comes before everything
// below it, and below the lowest common ancestor the near side comes first.  Same single forward scan
// over the insertion order as kd_first_of_two.  (The result list is filled at its head, kdtree.c:759-777:
// the reference's FIRST hit is the one reached LAST.)
inline int kd_range_first_of_two(const float *px, const float *py, int K, float qx, float qy, int A, int B) {
  if (A == B) return A;
  const float q[2] = {qx, qy};
  auto coord = [&](int n, int ax) { return ax == 0 ? px[n] : py[n]; };
  float lo[2] = {-INFINITY, -INFINITY}, hi[2] = {INFINITY, INFINITY};
  int cur = 0, axis = 0;
  for (;;) {
    if (cur == A || cur == B) return cur;
    const float split = coord(cur, axis);
    const bool near_is_left = (q[axis] - split) <= 0;
    const bool a_left = coord(A, axis) < split, b_left = coord(B, axis) < split;
    if (a_left != b_left) return (a_left == near_is_left) ? A : B;
    if (a_left)
      hi[axis] = split;
    else
      lo[axis] = split;
    axis ^= 1;
    int n = cur + 1;
    for (; n < K; ++n)
      if (px[n] >= lo[0] && px[n] < hi[0] && py[n] >= lo[1] && py[n] < hi[1]) break;
    if (n >= K) return A < B ? A : B;  // cannot happen: A and B are inside the box
    cur = n;
  }
}

// winner among a set of nodes that all have the minimal squared distance to (qx, qy)
inline int kd_tie_winner(const float *px, const float *py, int K, float qx, float qy,
                         const std::vector<int> &tied) {
  int w = tied[0];
  for (int n : tied)
    if (n == 0) return 0;  // the root is the initial best and equal distances never replace it
  for (size_t i = 1; i < tied.size(); ++i) w = kd_first_of_two(px, py, K, qx, qy, w, tied[i]);
  return w;
}

// Uniform grid over the node positions; cells hold singly linked lists of node slots.
class NodeGrid {
 public:
  void reset(float x0, float y0, float x1, float y1, float cell) {
    g_ = cell;
    inv_ = 1.0f / cell;
    x0_ = x0 - 2 * cell;
    y0_ = y0 - 2 * cell;
    W_ = (int)std::floor((x1 + 2 * cell - x0_) * inv_) + 1;
    H_ = (int)std::floor((y1 + 2 * cell - y0_) * inv_) + 1;
    if (W_ < 1) W_ = 1;
    if (H_ < 1) H_ = 1;
    head_.assign((size_t)W_ * H_, -1);
    next_.clear();
    px_.clear();
    py_.clear();
  }
  bool ready() const { return !head_.empty(); }
  size_t size() const { return px_.size(); }
  const float *xs() const { return px_.data(); }
  const float *ys() const { return py_.data(); }

  // every slot whose fp32 squared distance to (qx, qy) equals d2 (call after nearest() said tie)
  void tied_set(float qx, float qy, float d2, std::vector<int> &out) const {
    out.clear();
    const float rad = std::sqrt(d2) * 1.001f + 1e-6f;
    const int cx0 = clampi((int)std::floor((qx - rad - x0_) * inv_), 0, W_ - 1);
    const int cx1 = clampi((int)std::floor((qx + rad - x0_) * inv_), 0, W_ - 1);
    const int cy0 = clampi((int)std::floor((qy - rad - y0_) * inv_), 0, H_ - 1);
    const int cy1 = clampi((int)std::floor((qy + rad - y0_) * inv_), 0, H_ - 1);
    for (int yy = cy0; yy <= cy1; ++yy)
      for (int xx = cx0; xx <= cx1; ++xx)
        for (int s = head_[(size_t)yy * W_ + xx]; s >= 0; s = next_[s]) {
          float dd = 0;
          dd += (px_[s] - qx) * (px_[s] - qx);
          dd += (py_[s] - qy) * (py_[s] - qy);
          if (dd == d2) out.push_back(s);
        }
    std::sort(out.begin(), out.end());
  }
  float dist2(int s, float qx, float qy) const {
    float dd = 0;
    dd += (px_[s] - qx) * (px_[s] - qx);
    dd += (py_[s] - qy) * (py_[s] - qy);
    return dd;
  }
  // Is any slot within r of (qx, qy), as kd_nearest_range2 would report (d2 <= r * r, kdtree.c:270-301)?
  // 1 yes, 0 no, -1 cannot say without the tree: a slot within rounding of the radius may or may not be
  // reached by the tree walk (it prunes the far side of a split with |dx| >= r), so the caller asks the
  // tree replica in that (practically never occurring) case.
  int within(float qx, float qy, float r) const {
    const float r2 = r * r;
    const float lo = r2 * (1.0f - 4e-6f), hi = r2 * (1.0f + 4e-6f);
    const float rad = r * 1.001f + 1e-6f;
    const int cx0 = clampi((int)std::floor((qx - rad - x0_) * inv_), 0, W_ - 1);
    const int cx1 = clampi((int)std::floor((qx + rad - x0_) * inv_), 0, W_ - 1);
    const int cy0 = clampi((int)std::floor((qy - rad - y0_) * inv_), 0, H_ - 1);
    const int cy1 = clampi((int)std::floor((qy + rad - y0_) * inv_), 0, H_ - 1);
    bool doubt = false;
    for (int yy = cy0; yy <= cy1; ++yy)
      for (int xx = cx0; xx <= cx1; ++xx)
        for (int s = head_[(size_t)yy * W_ + xx]; s >= 0; s = next_[s]) {
          const float dd = dist2(s, qx, qy);
          if (dd < lo) return 1;
          if (dd <= hi) doubt = true;
        }
    return doubt ? -1 : 0;
  }

  // The slots kd_nearest_range2 would report around (qx, qy) (d2 <= r * r), in no particular order.
  // *doubt: some slot lies within rounding of the radius -- whether the tree walk reaches it depends on
  // the splits above it (the far side of a split with |dx| >= r is pruned), so the caller asks the tree.
  void range_set(float qx, float qy, float r, std::vector<int> &out, bool *doubt) const {
    out.clear();
    *doubt = false;
    const float r2 = r * r;
    const float lo = r2 * (1.0f - 4e-6f), hi = r2 * (1.0f + 4e-6f);
    const float rad = r * 1.001f + 1e-6f;
    const int cx0 = clampi((int)std::floor((qx - rad - x0_) * inv_), 0, W_ - 1);
    const int cx1 = clampi((int)std::floor((qx + rad - x0_) * inv_), 0, W_ - 1);
    const int cy0 = clampi((int)std::floor((qy - rad - y0_) * inv_), 0, H_ - 1);
    const int cy1 = clampi((int)std::floor((qy + rad - y0_) * inv_), 0, H_ - 1);
    for (int yy = cy0; yy <= cy1; ++yy)
      for (int xx = cx0; xx <= cx1; ++xx)
        for (int s = head_[(size_t)yy * W_ + xx]; s >= 0; s = next_[s]) {
          const float dd = dist2(s, qx, qy);
          if (dd < lo)
            out.push_back(s);
          else if (dd <= hi)
            *doubt = true;
        }
  }

  void insert(float x, float y) {  // slot index == insertion order
    const int slot = (int)px_.size();
    px_.push_back(x);
    py_.push_back(y);
    const size_t c = cell_of(x, y);
    next_.push_back(head_[c]);
    head_[c] = slot;
  }

  // nearest slot by the reference's fp32 squared distance; *tie is set when another slot has
  // exactly the same squared distance (then the kd-tree traversal order decides, not us)
  int nearest(float qx, float qy, bool *tie) const {
    *tie = false;
    if (px_.empty()) return -1;
    const int cx = clampi((int)std::floor((qx - x0_) * inv_), 0, W_ - 1);
    const int cy = clampi((int)std::floor((qy - y0_) * inv_), 0, H_ - 1);
    int best = -1;
    float best_d2 = 0;
    const int kmax = (W_ > H_ ? W_ : H_);
    for (int k = 0; k <= kmax; ++k) {
      // ring k of cells around (cx, cy)
      const int xa = cx - k, xb = cx + k, ya = cy - k, yb = cy + k;
      for (int yy = ya; yy <= yb; ++yy) {
        if (yy < 0 || yy >= H_) continue;
        const bool edge_row = (yy == ya || yy == yb);
        const int step = edge_row ? 1 : (xb - xa);
        for (int xx = xa; xx <= xb; xx += (step > 0 ? step : 1)) {
          if (xx < 0 || xx >= W_) continue;
          for (int s = head_[(size_t)yy * W_ + xx]; s >= 0; s = next_[s]) {
            float d2 = 0;
            d2 += (px_[s] - qx) * (px_[s] - qx);
            d2 += (py_[s] - qy) * (py_[s] - qy);
            if (best < 0 || d2 < best_d2) {
              best = s;
              best_d2 = d2;
              *tie = false;
            } else if (d2 == best_d2 && s != best) {
              *tie = true;
            }
          }
        }
      }
      if (best >= 0) {
        // the scanned block reaches at least k cells past the query's own cell on every side,
        // so every slot outside it is at least k*g away (1% slack for fp32 cell rounding)
        const float safe = (float)k * g_ * 0.99f;
        if (safe > 0 && best_d2 < safe * safe) break;
        if (xa <= 0 && ya <= 0 && xb >= W_ - 1 && yb >= H_ - 1) break;
      } else if (xa <= 0 && ya <= 0 && xb >= W_ - 1 && yb >= H_ - 1) {
        break;
      }
    }
    return best;
  }

 private:
  static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
  size_t cell_of(float x, float y) const {
    const int cx = clampi((int)std::floor((x - x0_) * inv_), 0, W_ - 1);
    const int cy = clampi((int)std::floor((y - y0_) * inv_), 0, H_ - 1);
    return (size_t)cy * W_ + cx;
  }
  float g_ = 1, inv_ = 1, x0_ = 0, y0_ = 0;
  int W_ = 0, H_ = 0;
  std::vector<int> head_, next_;
  std::vector<float> px_, py_;
};

}  // namespace trg
