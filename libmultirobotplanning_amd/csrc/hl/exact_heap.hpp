// Host-side array heap that replays boost::heap::d_ary_heap<arity<2>, mutable_<true>> operation by operation
// (push / pop / erase(handle) / ordered iteration), used for the high-level open and focal lists of CBS / ECBS
// (cbs.hpp:110-115,120,124,163; ecbs.hpp:143-145,172-190,227,232-233,276-279).  Which of several equal-cost CT nodes
// is expanded first is decided by this layout, so it has to match the reference's heap exactly.
// Rules (Boost.Heap d_ary_heap.hpp / detail/mutable_heap.hpp): push = append + sift-up while less(parent, child);
// pop = move last to root + sift-down preferring the FIRST maximal child and descending while !less(child, node);
// erase = swap up to the root unconditionally, then pop; ordered iteration = best-first traversal driven by a
// std::priority_queue with the same comparator.
#pragma once
#include <cstdint>
#include <queue>
#include <vector>

namespace mrp_hl {

// Keys are compared through `Less(idA, idB)` on element ids; the heap stores ids only.
template <typename Less>
class ExactHeap {
 public:
  explicit ExactHeap(Less less) : less_(less) {}

  bool empty() const { return slots_.empty(); }
  size_t size() const { return slots_.size(); }
  int32_t top() const { return slots_.front(); }

  void push(int32_t id) {
    if (static_cast<size_t>(id) >= where_.size()) where_.resize(id + 1, -1);
    slots_.push_back(id);
    where_[id] = static_cast<int32_t>(slots_.size()) - 1;
    up(slots_.size() - 1);
  }

  void pop() { removeRoot(); }

  void erase(int32_t id) {
    size_t i = static_cast<size_t>(where_[id]);
    while (i != 0) {
      size_t p = (i - 1) / 2;
      exchange(p, i);
      i = p;
    }
    removeRoot();
  }

  // boost increase(handle): the element's key got better — sift it up from where it is (a_star.hpp:143)
  void increase(int32_t id) { up(static_cast<size_t>(where_[id])); }

  // visit(id) -> false stops the walk
  template <typename Visit>
  void walkOrdered(Visit visit) const {
    if (slots_.empty()) return;
    auto cmp = [this](size_t a, size_t b) { return less_(slots_[a], slots_[b]); };
    std::priority_queue<size_t, std::vector<size_t>, decltype(cmp)> pending(cmp);
    size_t cur = 0;
    for (;;) {
      for (size_t c = 2 * cur + 1; c <= 2 * cur + 2 && c < slots_.size(); ++c) pending.push(c);
      if (!visit(slots_[cur])) return;
      if (pending.empty()) return;
      cur = pending.top();
      pending.pop();
    }
  }

 private:
  void exchange(size_t a, size_t b) {
    std::swap(slots_[a], slots_[b]);
    where_[slots_[a]] = static_cast<int32_t>(a);
    where_[slots_[b]] = static_cast<int32_t>(b);
  }
  void up(size_t i) {
    while (i != 0) {
      size_t p = (i - 1) / 2;
      if (!less_(slots_[p], slots_[i])) return;
      exchange(p, i);
      i = p;
    }
  }
  void down(size_t i) {
    const size_t n = slots_.size();
    for (;;) {
      size_t l = 2 * i + 1;
      if (l >= n) return;
      size_t best = l;
      if (l + 1 < n && less_(slots_[l], slots_[l + 1])) best = l + 1;
      if (less_(slots_[best], slots_[i])) return;
      exchange(best, i);
      i = best;
    }
  }
  void removeRoot() {
    size_t last = slots_.size() - 1;
    if (last != 0) exchange(0, last);
    where_[slots_.back()] = -1;
    slots_.pop_back();
    if (!slots_.empty()) down(0);
  }

  Less less_;
  std::vector<int32_t> slots_;
  std::vector<int32_t> where_;
};

}  // namespace mrp_hl
