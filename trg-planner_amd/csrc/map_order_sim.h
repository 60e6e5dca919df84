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
    size_ = 0;
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
    size_ = 0;
    rev_ = false;
  }
  // unordered_map::clear(): elements go, bucket array and policy stay
  void clear() {
    order_.clear();
    size_ = 0;
    rev_ = false;
  }
  // nodes[key] = ... for the next dense key (must equal size())
  void insert_next() { fill(1); }
  // the next n dense keys.  The policy is asked only where the real container asks with an effect
  // (_M_need_rehash returns at once while size + 1 <= _M_next_resize), so a fill costs one call per
  // rehash, and the keys between two rehashes are kept as one run.
  void fill(std::size_t n) {
    const std::size_t target = size_ + n;
    while (size_ < target) {
      if (size_ + 1 > policy_._M_next_resize) {
        const auto r = policy_._M_need_rehash(buckets_, size_, 1);
        if (r.first) {
          buckets_ = r.second;
          rev_ = !rev_;  // _M_rehash_aux: every element re-inserted at the front
        }
      }
      // keys [size_, end) arrive without a further rehash
      std::size_t end = policy_._M_next_resize > size_ ? policy_._M_next_resize : size_ + 1;
      if (end > target) end = target;
      const Run run{(int)size_, (int)end, !rev_};
      if (rev_) {
        if (!order_.empty() && !order_.back().front && order_.back().hi == run.lo)
          order_.back().hi = run.hi;
        else
          order_.push_back(run);
      } else {
        if (!order_.empty() && order_.front().front && order_.front().hi == run.lo)
          order_.front().hi = run.hi;
        else
          order_.push_front(run);
      }
      size_ = end;
    }
  }
  std::size_t size() const { return size_; }
  std::size_t bucket_count() const { return buckets_; }
  // keys in iteration order (begin() .. end())
  void iteration_order(std::vector<int> &out) const {
    out.clear();
    out.reserve(size_);
    // the element list: runs inserted at the front read newest first, runs appended at the back
    // oldest first; after an odd number of rehashes the whole list is read backwards
    auto emit = [&](const Run &r, bool backwards) {
      const bool descending = r.front != backwards;
      if (descending)
        for (int k = r.hi - 1; k >= r.lo; --k) out.push_back(k);
      else
        for (int k = r.lo; k < r.hi; ++k) out.push_back(k);
    };
    if (rev_) {
      for (auto it = order_.rbegin(); it != order_.rend(); ++it) emit(*it, true);
    } else {
      for (const Run &r : order_) emit(r, false);
    }
  }
  // `*this = other` of the real containers: bucket count, policy state and element order are
  // taken over (hashtable copy assignment, _M_assign_elements)
  void assign_from(const MapOrderSim &o) { *this = o; }

 private:
  struct Run {  // keys [lo, hi) inserted without a rehash in between; front: each went to the list's front
    int lo, hi;
    bool front;
  };
  std::size_t buckets_ = 1;
  std::__detail::_Prime_rehash_policy policy_;
  std::deque<Run> order_;
  std::size_t size_ = 0;
  bool rev_ = false;
};

}  // namespace trg
