// ORACLE — TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/load it.
//
// Restatement of the one third-party data structure the reference's hot path depends on:
//   boost::heap::d_ary_heap<T, boost::heap::arity<2>, boost::heap::mutable_<true>[, compare<C>]>
// Call sites in the reference: a_star.hpp:77,86,109,124,143 ; a_star_epsilon.hpp:102,107,122,136,
// 141-152,191,215-216,237,242,262,268 ; cbs.hpp:110-115,120,124,163 ; ecbs.hpp:143-145,172,177,227,
// 232-233,276,279.
//
// Boost is NOT vendored under /root/reference and is not installed in this image, so the reference
// cannot be built here (DESIGN.md "oracle pinning"). What is restated below is the published algorithm
// of Boost.Heap (boost/heap/d_ary_heap.hpp + boost/heap/detail/mutable_heap.hpp, Boost >= 1.58 as
// required by the reference's CMakeLists.txt:5):
//   * array binary max-heap w.r.t. cmp ("less"): top() is the element no other element is greater than;
//   * push      : append, sift-up while cmp(parent, child);
//   * pop       : swap(front, back), drop back, sift-down from the root;
//   * sift-down : pick std::max_element of the children (FIRST maximal child wins ties) and swap while
//                 !cmp(child, node)  -- i.e. the node also moves down on equality;
//   * increase  : sift-up;   update: sift-up if cmp(parent,node) else sift-down;
//   * erase(h)  : swap the element up to the root UNCONDITIONALLY, then pop();
//   * ordered iteration (ordered_begin/ordered_end): best-first traversal of the implicit tree driven by a
//     std::priority_queue of element references with the same comparator; children of the element just
//     yielded are pushed in index order.  The tie order therefore depends on libstdc++'s
//     std::push_heap/std::pop_heap, which we use directly (and restate explicitly in
//     OrderedWalkExplicit for the device kernels; tests check both agree).
// The heap layout decides every tie-break of the searches, hence the ECBS results.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <queue>
#include <vector>

namespace oracle {

// Elements live in a pool owned by the heap; a handle is the pool index (stable for the element's life).
template <typename T, typename Cmp = std::less<T>>
class MutableBinaryHeap {
 public:
  typedef std::size_t handle_type;
  static constexpr handle_type npos = static_cast<handle_type>(-1);

  explicit MutableBinaryHeap(const Cmp& cmp = Cmp()) : m_cmp(cmp) {}

  bool empty() const { return m_heap.empty(); }
  std::size_t size() const { return m_heap.size(); }

  T& operator[](handle_type h) { return m_pool[h]; }
  const T& operator[](handle_type h) const { return m_pool[h]; }

  handle_type push(const T& v) {
    handle_type h;
    if (!m_free.empty()) {
      h = m_free.back();
      m_free.pop_back();
      m_pool[h] = v;
    } else {
      h = m_pool.size();
      m_pool.push_back(v);
      m_pos.push_back(0);
    }
    m_heap.push_back(h);
    m_pos[h] = m_heap.size() - 1;
    siftUp(m_heap.size() - 1);
    return h;
  }

  const T& top() const { return m_pool[m_heap.front()]; }
  handle_type topHandle() const { return m_heap.front(); }

  void pop() {
    handle_type victim = m_heap.front();
    popArray();
    m_free.push_back(victim);
  }

  void erase(handle_type h) {
    std::size_t idx = m_pos[h];
    while (idx != 0) {  // unconditional bubble to the root
      std::size_t parent = (idx - 1) / 2;
      swapSlots(parent, idx);
      idx = parent;
    }
    popArray();
    m_free.push_back(h);
  }

  void increase(handle_type h) { siftUp(m_pos[h]); }

  void update(handle_type h) {
    std::size_t idx = m_pos[h];
    if (idx == 0) {
      siftDown(0);
      return;
    }
    std::size_t parent = (idx - 1) / 2;
    if (less(m_heap[parent], m_heap[idx]))
      siftUp(idx);
    else
      siftDown(idx);
  }

  // Best-first walk; fn(handle) returns false to stop. Uses the real std::priority_queue.
  template <typename Fn>
  void orderedWalk(Fn fn) const {
    if (m_heap.empty()) return;
    auto pqLess = [this](std::size_t a, std::size_t b) { return less(m_heap[a], m_heap[b]); };
    std::priority_queue<std::size_t, std::vector<std::size_t>, decltype(pqLess)> unvisited(pqLess);
    std::size_t cur = 0;
    for (;;) {
      // discover children of the current element (index order)
      std::size_t first = 2 * cur + 1;
      if (first < m_heap.size()) {
        std::size_t last = std::min(first + 1, m_heap.size() - 1);
        for (std::size_t i = first; i <= last; ++i) unvisited.push(i);
      }
      if (!fn(m_heap[cur])) return;
      if (unvisited.empty()) return;
      cur = unvisited.top();
      unvisited.pop();
    }
  }

  // Same walk with std::push_heap / std::pop_heap restated explicitly (libstdc++ bits/stl_heap.h
  // __push_heap / __adjust_heap); this is the form the HIP kernels implement.
  template <typename Fn>
  void orderedWalkExplicit(Fn fn) const {
    if (m_heap.empty()) return;
    std::vector<std::size_t> pq;  // array of heap-array indices
    auto lt = [this](std::size_t a, std::size_t b) { return less(m_heap[a], m_heap[b]); };
    auto pushHeap = [&](std::size_t value) {
      pq.push_back(value);
      std::ptrdiff_t hole = static_cast<std::ptrdiff_t>(pq.size()) - 1;
      std::ptrdiff_t parent = (hole - 1) / 2;
      while (hole > 0 && lt(pq[parent], value)) {
        pq[hole] = pq[parent];
        hole = parent;
        parent = (hole - 1) / 2;
      }
      pq[hole] = value;
    };
    auto popHeap = [&]() -> std::size_t {
      std::size_t result = pq.front();
      std::size_t value = pq.back();
      pq.pop_back();
      std::ptrdiff_t len = static_cast<std::ptrdiff_t>(pq.size());
      if (len == 0) return result;
      std::ptrdiff_t hole = 0, child = 0;
      while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (lt(pq[child], pq[child - 1])) child--;
        pq[hole] = pq[child];
        hole = child;
      }
      if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        pq[hole] = pq[child - 1];
        hole = child - 1;
      }
      std::ptrdiff_t parent = (hole - 1) / 2;
      while (hole > 0 && lt(pq[parent], value)) {
        pq[hole] = pq[parent];
        hole = parent;
        parent = (hole - 1) / 2;
      }
      pq[hole] = value;
      return result;
    };
    std::size_t cur = 0;
    for (;;) {
      std::size_t first = 2 * cur + 1;
      if (first < m_heap.size()) {
        std::size_t last = std::min(first + 1, m_heap.size() - 1);
        for (std::size_t i = first; i <= last; ++i) pushHeap(i);
      }
      if (!fn(m_heap[cur])) return;
      if (pq.empty()) return;
      cur = popHeap();
    }
  }

  // introspection for tests
  const std::vector<handle_type>& array() const { return m_heap; }

 private:
  bool less(handle_type a, handle_type b) const { return m_cmp(m_pool[a], m_pool[b]); }

  void swapSlots(std::size_t i, std::size_t j) {
    std::swap(m_heap[i], m_heap[j]);
    m_pos[m_heap[i]] = i;
    m_pos[m_heap[j]] = j;
  }

  void siftUp(std::size_t idx) {
    while (idx != 0) {
      std::size_t parent = (idx - 1) / 2;
      if (less(m_heap[parent], m_heap[idx])) {
        swapSlots(parent, idx);
        idx = parent;
      } else {
        return;
      }
    }
  }

  void siftDown(std::size_t idx) {
    const std::size_t n = m_heap.size();
    for (;;) {
      std::size_t first = 2 * idx + 1;
      if (first >= n) return;
      std::size_t best = first;  // std::max_element: first maximal element wins
      if (first + 1 < n && less(m_heap[best], m_heap[first + 1])) best = first + 1;
      if (!less(m_heap[best], m_heap[idx])) {
        swapSlots(best, idx);
        idx = best;
      } else {
        return;
      }
    }
  }

  void popArray() {
    std::size_t lastIdx = m_heap.size() - 1;
    if (lastIdx != 0) swapSlots(0, lastIdx);
    m_heap.pop_back();
    if (m_heap.empty()) return;
    siftDown(0);
  }

  Cmp m_cmp;
  std::vector<T> m_pool;
  std::vector<std::size_t> m_pos;      // handle -> index in m_heap
  std::vector<handle_type> m_heap;     // implicit binary tree of handles
  std::vector<handle_type> m_free;
};

}  // namespace oracle
