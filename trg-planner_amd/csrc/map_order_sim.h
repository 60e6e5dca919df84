// map_order_sim.h -- O(n) replica of the ITERATION ORDER of libstdc++'s
// std::unordered_map<int, T> for the one usage pattern cleanGraph depends on.
//
// TRG::cleanGraph renumbers nodes in the iteration order of trgStruct::nodes (trg.cpp:497-504), an
// unordered_map<int, Node*> that was filled with the dense keys 0, 1, 2, ... in creation order
// (trg.cpp:248).  With std::hash<int> being the identity and dense keys, every key lands in its own
// bucket, so libstdc++'s hashtable (bits/hashtable.h, _M_insert_bucket_begin / _M_rehash_aux)
// behaves like this: an insert goes to the FRONT of the element list, a rehash re-inserts the list
// front to back -- i.e. reverses it.  When a rehash happens is decided by the real
// std::__detail::_Prime_rehash_policy, used here directly, so bucket counts (which persist across
// clear() and are taken over by copy assignment) evolve exactly as in the reference's container.
// tests/cpp/map_order_check.cpp compares this replica with the real container.
#pragma once
#include <cstddef>
#include <deque>
#include <unordered_map>
#include <vector>

namespace trg {

class MapOrderSim {
 public:
  // state of a default-constructed, never used map
  MapOrderSim() { reset_pristine(); }

  void reset_pristine() {
    buckets_ = 1;
    policy_ = std::__detail::_Prime_rehash_policy();
    order_.clear();
    rev_ = false;
  }
  // adopt the bucket state of a real map (its elements are NOT copied; call clear()/fill next)
  template <typename M>
  void adopt_bucket_state(const M &real) {
    buckets_ = real.bucket_count();
    policy_ = std::__detail::_Prime_rehash_policy(real.max_load_factor());
    // _M_next_resize is 0 only for a map that never allocated (single bucket); afterwards it is
    // floor(bucket_count * max_load_factor)
    if (buckets_ > 1) policy_._M_next_bkt(buckets_);  // sets _M_next_resize for this count
    order_.clear();
    rev_ = false;
  }
  // unordered_map::clear(): elements go, bucket array and policy stay
  void clear() {
    order_.clear();
    rev_ = false;
  }
  // nodes[key] = ... for the next dense key (must equal size())
  void insert_next() {
    const std::size_t key = order_.size();
    const auto r = policy_._M_need_rehash(buckets_, order_.size(), 1);
    if (r.first) {
      buckets_ = r.second;
      rev_ = !rev_;  // _M_rehash_aux: every element re-inserted at the front
    }
    if (rev_) {
      order_.push_back((int)key);
    } else {
      order_.push_front((int)key);
    }
  }
  void fill(std::size_t n) {
    for (std::size_t i = 0; i < n; ++i) insert_next();
  }
  std::size_t size() const { return order_.size(); }
  std::size_t bucket_count() const { return buckets_; }
  // keys in iteration order (begin() .. end())
  void iteration_order(std::vector<int> &out) const {
    out.clear();
    out.reserve(order_.size());
    if (rev_) {
      for (auto it = order_.rbegin(); it != order_.rend(); ++it) out.push_back(*it);
    } else {
      for (int k : order_) out.push_back(k);
    }
  }
  // `*this = other` of the real containers: bucket count, policy state and element order are
  // taken over (hashtable copy assignment, _M_assign_elements)
  void assign_from(const MapOrderSim &o) { *this = o; }

 private:
  std::size_t buckets_ = 1;
  std::__detail::_Prime_rehash_policy policy_;
  std::deque<int> order_;
  bool rev_ = false;
};

}  // namespace trg
