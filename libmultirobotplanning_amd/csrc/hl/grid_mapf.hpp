// High-level side of the grid MAPF domain (example/ecbs.cpp, example/cbs.cpp): conflict-tree node data, first-conflict
// detection, constraint creation and the CT-node focal heuristic.  Paths are immutable and shared between CT nodes
// (a child replaces exactly one agent's path), instead of the reference's deep copy per child (ecbs.hpp:253).
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <vector>

namespace mrp_hl {

struct Path {                 // PlanResult of one agent (planresult.hpp:18-27); state k is at time k, every step costs 1
  std::vector<int32_t> xy;    // [len][2]
  int32_t cost = 0;
  int32_t fmin = 0;
  int32_t len() const { return static_cast<int32_t>(xy.size() / 2); }
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

struct CTNode {               // HighLevelNode (cbs.hpp:175-207, ecbs.hpp:308-342)
  std::vector<PathPtr> solution;
  std::vector<ConsPtr> constraints;
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

inline int32_t maxT(const std::vector<PathPtr>& sol) {
  int32_t m = 0;
  for (const auto& p : sol) m = std::max<int32_t>(m, p->len() - 1);
  return m;
}

// getFirstConflict (ecbs.cpp:401-452): scan order is t ascending; at each t all vertex pairs (i<j) before all
// swap pairs (i<j); the final time step is never checked (t < max_t).
inline bool firstConflict(const std::vector<PathPtr>& sol, Conflict& out, std::vector<int32_t>& scratch) {
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

// focalHeuristic (ecbs.cpp:315-350): number of vertex + swap conflicts over all pairs and t < max_t
inline int32_t countConflicts(const std::vector<PathPtr>& sol, std::vector<int32_t>& scratch) {
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

// Conflicts (vertex + swap, t < T) between path `p` of agent `ag` and every other agent of `sol` — the terms of
// focalHeuristic (ecbs.cpp:315-350) that involve agent `ag`.  A CT child differs from its parent in one agent's path,
// so while the scan horizon T = max_t is unchanged its focal heuristic is
//     parent's value - conflictsOfAgent(parent's path of ag) + conflictsOfAgent(new path of ag)
// which is O(T*N) instead of the reference's O(T*N^2); the integers are the same (sum over the same pairs and steps).
inline int32_t conflictsOfAgent(const std::vector<PathPtr>& sol, int32_t ag, const Path& p, int32_t T) {
  const int32_t n = static_cast<int32_t>(sol.size());
  int32_t total = 0;
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
