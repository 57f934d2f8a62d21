// ORACLE — TEST INFRASTRUCTURE ONLY (see heap_restated.hpp header).
// Config 1 plumbing: the timeless 2-D grid Environment of example/a_star.cpp:15-127 run through the restated
// AStar (a_star.hpp:63-161).  Neighbour order Up, Down, Left, Right (a_star.cpp:77-100).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "search_restated.hpp"

namespace oracle {
namespace grid2d {

struct State {
  int x, y;
  bool operator==(const State& o) const { return x == o.x && y == o.y; }
};
struct StateHash {
  std::size_t operator()(const State& s) const {
    std::size_t seed = 0;
    seed ^= std::hash<int>()(s.x) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    seed ^= std::hash<int>()(s.y) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    return seed;
  }
};
enum class Action { Up, Down, Left, Right };

class Environment {
 public:
  Environment(int dimx, int dimy, const uint8_t* mask, State goal)
      : m_dimx(dimx), m_dimy(dimy), m_mask(mask), m_goal(goal) {}
  int admissibleHeuristic(const State& s) { return std::abs(s.x - m_goal.x) + std::abs(s.y - m_goal.y); }
  bool isSolution(const State& s) { return s == m_goal; }
  void getNeighbors(const State& s, std::vector<Neighbor<State, Action, int>>& out) {
    out.clear();
    const State cand[4] = {{s.x, s.y + 1}, {s.x, s.y - 1}, {s.x - 1, s.y}, {s.x + 1, s.y}};
    const Action act[4] = {Action::Up, Action::Down, Action::Left, Action::Right};
    for (int k = 0; k < 4; ++k)
      if (stateValid(cand[k])) out.emplace_back(cand[k], act[k], 1);
  }
  void onExpandNode(const State&, int, int) { ++expanded; }
  void onDiscover(const State&, int, int) {}
  bool stateValid(const State& s) const {
    return s.x >= 0 && s.x < m_dimx && s.y >= 0 && s.y < m_dimy && !m_mask[s.y * m_dimx + s.x];
  }
  int64_t expanded = 0;

 private:
  int m_dimx, m_dimy;
  const uint8_t* m_mask;
  State m_goal;
};

// mirrors main(): search only if the start is valid (a_star.cpp:192-194); returns #states, 0 on failure
inline int solve(int dimx, int dimy, const uint8_t* mask, int sx, int sy, int gx, int gy, int32_t* statesXY, int cap,
                 int64_t* expanded) {
  Environment env(dimx, dimy, mask, State{gx, gy});
  AStar<State, Action, int, Environment, StateHash> astar(env);
  PlanResult<State, Action, int> sol;
  bool ok = false;
  State start{sx, sy};
  if (env.stateValid(start)) ok = astar.search(start, sol);
  if (expanded) *expanded = env.expanded;
  if (!ok) return 0;
  int n = static_cast<int>(sol.states.size());
  for (int i = 0; i < n && i < cap; ++i) {
    statesXY[2 * i] = sol.states[i].first.x;
    statesXY[2 * i + 1] = sol.states[i].first.y;
  }
  return n;
}

}  // namespace grid2d
}  // namespace oracle
