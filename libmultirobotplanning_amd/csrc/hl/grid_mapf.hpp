// High-level side of the grid MAPF domain (example/ecbs.cpp, example/cbs.cpp): conflict-tree node data, first-conflict
// detection, constraint creation and the CT-node focal heuristic.  Paths are immutable and shared between CT nodes
// (a child replaces exactly one agent's path), instead of the reference's deep copy per child (ecbs.hpp:253).
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <vector>

namespace mrp_hl {

// Slot ids of the engine's device-resident path store (mrp_ll_path_store_reserve), handed out per worker thread.
struct SlotPool {
  std::vector<int32_t> freed;
  int32_t next = 0, cap = 0;  // slots next .. cap - 1 have never been handed out (a pool may own a sub-range of the store)
  int32_t take() {
    if (!freed.empty()) {
      const int32_t s = freed.back();
      freed.pop_back();
      return s;
    }
    return next < cap ? next++ : -1;
  }
  void give(int32_t s) {
    if (s >= 0) freed.push_back(s);
  }
};

struct Path {                 // PlanResult of one agent (planresult.hpp:18-27); state k is at time k, every step costs 1
  std::vector<int32_t> xy;    // [len][2]
  int32_t cost = 0;
  int32_t fmin = 0;
  bool fits8 = false;         // every coordinate is in 0..255 (set by whoever fills xy; enables the linear scans)
  std::vector<uint16_t> cell; // fits8 only: x | y << 8 per state (packCells), what the vectorised conflict counts read
  // the search that produced this path also left it in the engine's device path store; the slot goes back
  // to its pool with the last conflict-tree node that shares the path (ecbs.hpp:253 would have copied it instead)
  int32_t devSlot = -1;
  SlotPool* pool = nullptr;
  int32_t len() const { return static_cast<int32_t>(xy.size() / 2); }
  void packCells() {
    cell.resize(xy.size() / 2);
    for (size_t k = 0; k < cell.size(); ++k) cell[k] = static_cast<uint16_t>(xy[2 * k] | (xy[2 * k + 1] << 8));
  }
  Path() = default;
  Path(const Path&) = delete;
  Path& operator=(const Path&) = delete;
  ~Path() {
    if (pool) pool->give(devSlot);
  }
};
typedef std::shared_ptr<const Path> PathPtr;

struct ConstraintSet {        // Constraints of one agent (ecbs.cpp:180-214) as flat arrays for the C-ABI
  std::vector<int32_t> vertex;  // [n][3] time, x, y
  std::vector<int32_t> edge;    // [n][5] time, x1, y1, x2, y2
};
typedef std::shared_ptr<const ConstraintSet> ConsPtr;

struct Conflict {             // ecbs.cpp:80-106
  enum Type { Vertex, Edge };
  int32_t time;
  int32_t agent1, agent2;
  Type type;
  int32_t x1, y1, x2, y2;
};

// The per-agent arrays of a conflict-tree node.  A child is its parent with ONE entry replaced (ecbs.hpp:253-263 copies
// all N PlanResults); a plain vector of shared_ptr still touches N reference counts — N cache lines — per copy and per
// destruction, which with 100 agents was most of a tree step.  Entries therefore live in shared chunks of 16: copying a
// node copies N / 16 chunk pointers, replacing an entry clones one chunk.
template <class P>
class CowVec {
 public:
  static constexpr size_t kChunk = 16;
  struct Chunk {
    P v[kChunk];
  };
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  const P& operator[](size_t i) const { return chunks_[i / kChunk]->v[i % kChunk]; }
  void assign(size_t n, const P& value) {
    n_ = n;
    chunks_.clear();
    if (n == 0) return;
    auto c = std::make_shared<Chunk>();
    for (size_t k = 0; k < kChunk; ++k) c->v[k] = value;
    chunks_.assign((n + kChunk - 1) / kChunk, c);  // chunks are immutable once shared: one chunk serves them all
  }
  void set(size_t i, const P& value) {
    std::shared_ptr<Chunk>& c = chunks_[i / kChunk];
    if (c.use_count() != 1) c = std::make_shared<Chunk>(*c);
    c->v[i % kChunk] = value;
  }
  class const_iterator {
   public:
    const_iterator(const CowVec* v, size_t i) : v_(v), i_(i) {}
    const P& operator*() const { return (*v_)[i_]; }
    const_iterator& operator++() {
      ++i_;
      return *this;
    }
    bool operator!=(const const_iterator& o) const { return i_ != o.i_; }

   private:
    const CowVec* v_;
    size_t i_;
  };
  const_iterator begin() const { return const_iterator(this, 0); }
  const_iterator end() const { return const_iterator(this, n_); }

 private:
  size_t n_ = 0;
  std::vector<std::shared_ptr<Chunk>> chunks_;
};
typedef CowVec<PathPtr> PathVec;
typedef CowVec<ConsPtr> ConsVec;

struct CTNode {               // HighLevelNode (cbs.hpp:175-207, ecbs.hpp:308-342)
  PathVec solution;
  ConsVec constraints;
  int32_t cost = 0;
  int32_t LB = 0;
  int32_t focalHeuristic = 0;
  int32_t id = 0;
};

// getState (ecbs.cpp:486-495): an agent past the end of its path stays on its last cell
inline void cellAt(const Path& p, int32_t t, int32_t& x, int32_t& y) {
  int32_t k = t < p.len() ? t : p.len() - 1;
  x = p.xy[2 * k];
  y = p.xy[2 * k + 1];
}

inline int32_t maxT(const PathVec& sol) {
  int32_t m = 0;
  for (const auto& p : sol) m = std::max<int32_t>(m, p->len() - 1);
  return m;
}

// Per-thread cell -> agent table for the linear-time scans below (coordinates fit 8 bits each: the engine's grids are
// at most 256 x 256).  Entries are validated by a stamp, so starting a new time step costs nothing.
struct CellTable {
  std::vector<uint32_t> stamp;
  std::vector<int32_t> who;
  uint32_t now = 0;
  CellTable() : stamp(65536, 0), who(65536, 0) {}
  void nextStep() {
    if (++now == 0) {
      std::fill(stamp.begin(), stamp.end(), 0u);
      now = 1;
    }
  }
  static CellTable& local() {
    static thread_local CellTable t;
    return t;
  }
};
inline bool fitsCellTable(const PathVec& sol) {
  for (const auto& p : sol)
    if (!p->fits8) return false;
  return true;
}

// getFirstConflict (ecbs.cpp:401-452): scan order is t ascending; at each t all vertex pairs (i<j) before all
// swap pairs (i<j); the final time step is never checked (t < max_t).  Quadratic restatement, kept for grids whose
// coordinates do not fit the cell table and as the cross-check of the linear scan in the CPU tests.
inline bool firstConflictQuadratic(const PathVec& sol, Conflict& out, std::vector<int32_t>& scratch) {
  const int32_t n = static_cast<int32_t>(sol.size());
  const int32_t T = maxT(sol);
  scratch.resize(static_cast<size_t>(n) * 4);
  int32_t* cur = scratch.data();            // x,y at t
  int32_t* nxt = scratch.data() + 2 * n;    // x,y at t+1
  for (int32_t i = 0; i < n; ++i) cellAt(*sol[i], 0, cur[2 * i], cur[2 * i + 1]);
  for (int32_t t = 0; t < T; ++t) {
    for (int32_t i = 0; i < n; ++i) cellAt(*sol[i], t + 1, nxt[2 * i], nxt[2 * i + 1]);
    for (int32_t i = 0; i < n; ++i)
      for (int32_t j = i + 1; j < n; ++j)
        if (cur[2 * i] == cur[2 * j] && cur[2 * i + 1] == cur[2 * j + 1]) {
          out = Conflict{t, i, j, Conflict::Vertex, cur[2 * i], cur[2 * i + 1], 0, 0};
          return true;
        }
    for (int32_t i = 0; i < n; ++i)
      for (int32_t j = i + 1; j < n; ++j)
        if (cur[2 * i] == nxt[2 * j] && cur[2 * i + 1] == nxt[2 * j + 1] && nxt[2 * i] == cur[2 * j] &&
            nxt[2 * i + 1] == cur[2 * j + 1]) {
          out = Conflict{t, i, j, Conflict::Edge, cur[2 * i], cur[2 * i + 1], nxt[2 * i], nxt[2 * i + 1]};
          return true;
        }
    std::swap(cur, nxt);
  }
  return false;
}

// The paths of a solution as plain arrays for the linear scans below: a scan looks at every agent at every time step, and
// going through the chunked vector, the shared pointer, the Path and its vector for each of those looks was most of the
// scan at a hundred agents (four dependent loads per look).  Packed cells (x | y << 8 per state, Path::packCells) are the
// scans' keys already, and a quarter of the bytes of the coordinate pairs.  false: some path has no packed cells.
struct PackedView {
  std::vector<const uint16_t*> c;
  std::vector<int32_t> len;
  static PackedView& local() {
    static thread_local PackedView v;
    return v;
  }
  bool gather(const PathVec& sol) {
    const size_t n = sol.size();
    c.resize(n);
    len.resize(n);
    size_t i = 0;
    for (const auto& p : sol) {
      const int32_t l = p->len();
      if (l == 0 || static_cast<int32_t>(p->cell.size()) != l) return false;
      c[i] = p->cell.data();
      len[i] = l;
      ++i;
    }
    return true;
  }
};

// The same result in O(T * N): at each t the agents are entered into the cell table in index order.
//  * vertex: the lexicographically first pair (i<j) on a common cell is (first occupant, second occupant) of some
//    cell; the minimum over cells is kept while scanning j upwards.
//  * swap: reached only when no two agents share a cell at t, so "the agent now standing on i's next cell" is unique;
//    scanning i upwards, the first i whose partner moves onto i's cell is the first pair in (i<j) order (a partner
//    j < i would have reported the pair when the scan was at j).
template <bool PACKED>
inline bool firstConflictLinear(const PathVec& sol, const PackedView& pv, Conflict& out, std::vector<int32_t>& scratch) {
  const int32_t n = static_cast<int32_t>(sol.size());
  CellTable& tab = CellTable::local();
  scratch.resize(static_cast<size_t>(n) * 2);
  int32_t* cur = scratch.data();        // y << 8 | x at t
  int32_t* nxt = scratch.data() + n;    // at t + 1
  auto keyAt = [&](int32_t i, int32_t t) -> int32_t {
    if constexpr (PACKED) {
      const int32_t l = pv.len[i];
      return pv.c[i][t < l ? t : l - 1];
    } else {
      const Path& p = *sol[i];
      const int32_t k = t < p.len() ? t : p.len() - 1;
      return (p.xy[2 * k + 1] << 8) | p.xy[2 * k];
    }
  };
  int32_t T = 0;
  if constexpr (PACKED) {
    for (int32_t i = 0; i < n; ++i) T = std::max(T, pv.len[i] - 1);
  } else {
    T = maxT(sol);
  }
  for (int32_t i = 0; i < n; ++i) cur[i] = keyAt(i, 0);
  for (int32_t t = 0; t < T; ++t) {
    for (int32_t i = 0; i < n; ++i) nxt[i] = keyAt(i, t + 1);
    tab.nextStep();
    int32_t bi = n, bj = n;
    for (int32_t j = 0; j < n; ++j) {
      const int32_t key = cur[j];
      if (tab.stamp[key] == tab.now) {
        const int32_t i = tab.who[key];
        if (i < bi || (i == bi && j < bj)) {
          bi = i;
          bj = j;
        }
      } else {
        tab.stamp[key] = tab.now;
        tab.who[key] = j;
      }
    }
    if (bi < n) {
      out = Conflict{t, bi, bj, Conflict::Vertex, cur[bi] & 255, cur[bi] >> 8, 0, 0};
      return true;
    }
    for (int32_t i = 0; i < n; ++i) {
      const int32_t key = nxt[i];
      if (tab.stamp[key] != tab.now) continue;
      const int32_t j = tab.who[key];
      if (j != i && nxt[j] == cur[i]) {
        out = Conflict{t, i, j, Conflict::Edge, cur[i] & 255, cur[i] >> 8, nxt[i] & 255, nxt[i] >> 8};
        return true;
      }
    }
    std::swap(cur, nxt);
  }
  return false;
}
inline bool firstConflict(const PathVec& sol, Conflict& out, std::vector<int32_t>& scratch) {
  PackedView& pv = PackedView::local();
  if (pv.gather(sol)) return firstConflictLinear<true>(sol, pv, out, scratch);  // (packed cells exist only when every coordinate fits 8 bits)
  if (!fitsCellTable(sol)) return firstConflictQuadratic(sol, out, scratch);
  return firstConflictLinear<false>(sol, pv, out, scratch);
}

// focalHeuristic (ecbs.cpp:315-350): number of vertex + swap conflicts over all pairs and t < max_t
inline int32_t countConflictsQuadratic(const PathVec& sol, std::vector<int32_t>& scratch) {
  const int32_t n = static_cast<int32_t>(sol.size());
  const int32_t T = maxT(sol);
  scratch.resize(static_cast<size_t>(n) * 4);
  int32_t* cur = scratch.data();
  int32_t* nxt = scratch.data() + 2 * n;
  for (int32_t i = 0; i < n; ++i) cellAt(*sol[i], 0, cur[2 * i], cur[2 * i + 1]);
  int32_t total = 0;
  for (int32_t t = 0; t < T; ++t) {
    for (int32_t i = 0; i < n; ++i) cellAt(*sol[i], t + 1, nxt[2 * i], nxt[2 * i + 1]);
    for (int32_t i = 0; i < n; ++i) {
      const int32_t cx = cur[2 * i], cy = cur[2 * i + 1], nx = nxt[2 * i], ny = nxt[2 * i + 1];
      for (int32_t j = i + 1; j < n; ++j) {
        total += (cx == cur[2 * j] && cy == cur[2 * j + 1]);
        total += (cx == nxt[2 * j] && cy == nxt[2 * j + 1] && nx == cur[2 * j] && ny == cur[2 * j + 1]);
      }
    }
    std::swap(cur, nxt);
  }
  return total;
}

// The same integer with the cell table: agents on one cell are chained (who = last entered, link[] = the one before),
// an agent entering a cell that already holds c agents adds c vertex conflicts, and the swap partners of agent i are
// the agents j > i now on i's next cell whose next cell is i's current one.
template <bool PACKED>
inline int32_t countConflictsLinear(const PathVec& sol, const PackedView& pv, std::vector<int32_t>& scratch) {
  const int32_t n = static_cast<int32_t>(sol.size());
  CellTable& tab = CellTable::local();
  scratch.resize(static_cast<size_t>(n) * 4);
  int32_t* cur = scratch.data();
  int32_t* nxt = scratch.data() + n;
  int32_t* link = scratch.data() + 2 * n;   // previous agent on the same cell, -1 = none
  int32_t* depth = scratch.data() + 3 * n;  // agents entered before this one on the same cell
  auto keyAt = [&](int32_t i, int32_t t) -> int32_t {
    if constexpr (PACKED) {
      const int32_t l = pv.len[i];
      return pv.c[i][t < l ? t : l - 1];
    } else {
      const Path& p = *sol[i];
      const int32_t k = t < p.len() ? t : p.len() - 1;
      return (p.xy[2 * k + 1] << 8) | p.xy[2 * k];
    }
  };
  int32_t T = 0;
  if constexpr (PACKED) {
    for (int32_t i = 0; i < n; ++i) T = std::max(T, pv.len[i] - 1);
  } else {
    T = maxT(sol);
  }
  for (int32_t i = 0; i < n; ++i) cur[i] = keyAt(i, 0);
  int32_t total = 0;
  for (int32_t t = 0; t < T; ++t) {
    for (int32_t i = 0; i < n; ++i) nxt[i] = keyAt(i, t + 1);
    tab.nextStep();
    for (int32_t j = 0; j < n; ++j) {
      const int32_t key = cur[j];
      if (tab.stamp[key] == tab.now) {
        link[j] = tab.who[key];
        depth[j] = depth[link[j]] + 1;
        total += depth[j];
      } else {
        tab.stamp[key] = tab.now;
        link[j] = -1;
        depth[j] = 0;
      }
      tab.who[key] = j;
    }
    for (int32_t i = 0; i < n; ++i) {
      const int32_t key = nxt[i];
      if (tab.stamp[key] != tab.now) continue;
      for (int32_t j = tab.who[key]; j > i; j = link[j]) total += (nxt[j] == cur[i]);  // chain is index-descending
    }
    std::swap(cur, nxt);
  }
  return total;
}
inline int32_t countConflicts(const PathVec& sol, std::vector<int32_t>& scratch) {
  PackedView& pv = PackedView::local();
  if (pv.gather(sol)) return countConflictsLinear<true>(sol, pv, scratch);
  if (!fitsCellTable(sol)) return countConflictsQuadratic(sol, scratch);
  return countConflictsLinear<false>(sol, pv, scratch);
}

// Conflicts (vertex + swap, t < T) between path `p` of agent `ag` and every other agent of `sol` — the terms of
// focalHeuristic (ecbs.cpp:315-350) that involve agent `ag`.  A CT child differs from its parent in one agent's path,
// so while the scan horizon T = max_t is unchanged its focal heuristic is
//     parent's value - conflictsOfAgent(parent's path of ag) + conflictsOfAgent(new path of ag)
// which is O(T*N) instead of the reference's O(T*N^2); the integers are the same (sum over the same pairs and steps).
inline int32_t conflictsOfAgent(const PathVec& sol, int32_t ag, const Path& p, int32_t T) {
  const int32_t n = static_cast<int32_t>(sol.size());
  int32_t total = 0;
  // With 50-100 agents this count (twice per conflict-tree child) was most of a worker thread's time.  When every path
  // carries its packed cells, agent p's cells are laid out once for t = 0..T (clamped as getState does) and each other
  // path is compared in two branch-free loops the compiler vectorises: while q still moves, and after q has stopped.
  bool packed = !p.cell.empty() && static_cast<int32_t>(p.cell.size()) == p.len();
  for (int32_t j = 0; j < n && packed; ++j)
    if (j != ag && (sol[j]->cell.empty() || static_cast<int32_t>(sol[j]->cell.size()) != sol[j]->len())) packed = false;
  if (packed && T > 0) {
    static thread_local std::vector<uint16_t> pc;
    pc.resize(static_cast<size_t>(T) + 2);
    const int32_t lp = p.len();
    for (int32_t t = 0; t <= T + 1; ++t) pc[t] = p.cell[t < lp ? t : lp - 1];
    const uint16_t* P0 = pc.data();
    for (int32_t j = 0; j < n; ++j) {
      if (j == ag) continue;
      const Path& q = *sol[j];
      const uint16_t* Q = q.cell.data();
      const int32_t lq = q.len();
      const int32_t m = std::min(T, lq - 1);  // t < m: q(t) and q(t+1) are both real states
      int32_t c = 0;
      for (int32_t t = 0; t < m; ++t) c += (P0[t] == Q[t]) + ((P0[t] == Q[t + 1]) & (P0[t + 1] == Q[t]));
      const uint16_t g = Q[lq - 1];           // from then on q stays on its last cell
      for (int32_t t = m; t < T; ++t) c += (P0[t] == g) + ((P0[t] == g) & (P0[t + 1] == g));
      total += c;
    }
    return total;
  }
  for (int32_t j = 0; j < n; ++j) {
    if (j == ag) continue;
    const Path& q = *sol[j];
    for (int32_t t = 0; t < T; ++t) {
      int32_t px, py, pnx, pny, qx, qy, qnx, qny;
      cellAt(p, t, px, py);
      cellAt(p, t + 1, pnx, pny);
      cellAt(q, t, qx, qy);
      cellAt(q, t + 1, qnx, qny);
      total += (px == qx && py == qy);
      total += (px == qnx && py == qny && pnx == qx && pny == qy);
    }
  }
  return total;
}

// conflictsOfAgent's packed form over a gathered view (the entry of agent `ag` itself is not looked at): what the commit of a
// conflict-tree child calls twice — for the new and for the replaced path — after ONE walk over the node's paths.
inline int32_t conflictsOfAgentPacked(const PackedView& pv, int32_t ag, const Path& p, int32_t T) {
  if (T <= 0) return 0;
  const int32_t n = static_cast<int32_t>(pv.c.size());
  static thread_local std::vector<uint16_t> pc;
  pc.resize(static_cast<size_t>(T) + 2);
  const int32_t lp = p.len();
  for (int32_t t = 0; t <= T + 1; ++t) pc[t] = p.cell[t < lp ? t : lp - 1];
  const uint16_t* P0 = pc.data();
  int32_t total = 0;
  for (int32_t j = 0; j < n; ++j) {
    if (j == ag) continue;
    const uint16_t* Q = pv.c[j];
    const int32_t lq = pv.len[j];
    const int32_t m = std::min(T, lq - 1);  // t < m: q(t) and q(t+1) are both real states
    int32_t c = 0;
    for (int32_t t = 0; t < m; ++t) c += (P0[t] == Q[t]) + ((P0[t] == Q[t + 1]) & (P0[t + 1] == Q[t]));
    const uint16_t g = Q[lq - 1];           // from then on q stays on its last cell
    for (int32_t t = m; t < T; ++t) c += (P0[t] == g) + ((P0[t] == g) & (P0[t + 1] == g));
    total += c;
  }
  return total;
}

// createConstraintsFromConflict (ecbs.cpp:454-472): returns the constraint added for (agent1, agent2); the map is
// iterated in ascending agent order (std::map, ecbs.hpp:247-249) and agent1 < agent2 always holds.
inline void splitConflict(const Conflict& c, ConstraintSet& forAgent1, ConstraintSet& forAgent2) {
  if (c.type == Conflict::Vertex) {
    forAgent1.vertex = {c.time, c.x1, c.y1};
    forAgent2.vertex = {c.time, c.x1, c.y1};
  } else {
    forAgent1.edge = {c.time, c.x1, c.y1, c.x2, c.y2};
    forAgent2.edge = {c.time, c.x2, c.y2, c.x1, c.y1};
  }
}

inline ConsPtr withAdded(const ConsPtr& base, const ConstraintSet& extra) {  // Constraints::add (ecbs.cpp:184-189)
  auto out = std::make_shared<ConstraintSet>();
  if (base) *out = *base;
  // unordered_set semantics: inserting an element that is already present changes nothing
  auto hasV = [&](const int32_t* v) {
    for (size_t k = 0; k < out->vertex.size(); k += 3)
      if (out->vertex[k] == v[0] && out->vertex[k + 1] == v[1] && out->vertex[k + 2] == v[2]) return true;
    return false;
  };
  auto hasE = [&](const int32_t* e) {
    for (size_t k = 0; k < out->edge.size(); k += 5)
      if (out->edge[k] == e[0] && out->edge[k + 1] == e[1] && out->edge[k + 2] == e[2] && out->edge[k + 3] == e[3] &&
          out->edge[k + 4] == e[4])
        return true;
    return false;
  };
  for (size_t k = 0; k < extra.vertex.size(); k += 3)
    if (!hasV(&extra.vertex[k])) out->vertex.insert(out->vertex.end(), &extra.vertex[k], &extra.vertex[k] + 3);
  for (size_t k = 0; k < extra.edge.size(); k += 5)
    if (!hasE(&extra.edge[k])) out->edge.insert(out->edge.end(), &extra.edge[k], &extra.edge[k] + 5);
  return out;
}

}  // namespace mrp_hl
