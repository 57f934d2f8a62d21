// ORACLE — TEST INFRASTRUCTURE ONLY (see heap_restated.hpp header).
//
// CPU restatement of the LOW-LEVEL side of the task-assignment callers: the grid Environment of
// example/cbs_ta.cpp as the low-level search sees it, on top of the restated AStar (a_star.hpp, search_restated.hpp).
//   setLowLevelContext        example/cbs_ta.cpp:283-303   (task may be nullptr: then EVERY vertex constraint delays the end)
//   admissibleHeuristic       :305-311   shortest-path table to the task's cell, 0 without a task
//   isSolution                :313-319   (at the task's cell, or anywhere without a task) and time > m_lastGoalConstraint
//   getNeighbors              :321-367   Wait, Left, Right, Up, Down; Wait costs 0 AT THE GOAL (always, without a task), else 1
//   stateValid / transitionValid  :483-496 (same as example/cbs.cpp), getState :472-481
//   ShortestPathHeuristic     example/shortest_path_heuristic.hpp:12-63: all-pairs shortest paths on the 4-connected free
//                             cells with unit weights (Boost floyd_warshall_all_pairs_shortest_paths) — restated as one
//                             breadth-first search per goal, which yields the same integers; an unreachable pair keeps
//                             Boost's "infinity" (std::numeric_limits<int>::max()).
// Because Wait can cost 0, g != time: a state (time, x, y) can be reached again with a smaller g and the decrease-key
// branch a_star.hpp:139-145 is live (it is dead for example/cbs.cpp's Environment).
// The conflict-tree side (cbs_ta.hpp:85-215) is restated only as far as a FIXED task assignment goes (cbsFixedTasks): the
// assignment solvers (assignment.hpp, next_best_assignment.hpp: Boost.Graph min-cost flow) are out of scope (SURVEY.md §2
// #12).  That is enough to pin this file against the reference's own known answers: test/test_cbs_ta.py:24-38 asserts the
// optimum over all assignments, which on those three fixtures a brute-force over the (at most two) assignments reproduces.
#pragma once
#include <climits>
#include <deque>
#include <map>

#include "mapf_restated.hpp"

namespace oracle {
namespace ta {

using mapf::Action;
using mapf::Cell;
using mapf::CellHash;
using mapf::Conflict;
using mapf::Constraints;
using mapf::EdgeConstraint;
using mapf::Plan;
using mapf::State;
using mapf::StateHash;
using mapf::VertexConstraint;

// shortest_path_heuristic.hpp: distance from every cell to `goal` (INT_MAX: unreachable / obstacle)
inline std::vector<int> shortestPathTable(int dimx, int dimy, const std::unordered_set<Cell, CellHash>& obstacles, Cell goal) {
  std::vector<int> dist(static_cast<std::size_t>(dimx) * dimy, INT_MAX);
  if (goal.x < 0 || goal.x >= dimx || goal.y < 0 || goal.y >= dimy || obstacles.count(goal)) return dist;
  std::deque<Cell> q;
  dist[goal.x + dimx * goal.y] = 0;
  q.push_back(goal);
  static const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
  while (!q.empty()) {
    Cell c = q.front();
    q.pop_front();
    for (int k = 0; k < 4; ++k) {
      Cell n{c.x + dx[k], c.y + dy[k]};
      if (n.x < 0 || n.x >= dimx || n.y < 0 || n.y >= dimy || obstacles.count(n)) continue;
      if (dist[n.x + dimx * n.y] != INT_MAX) continue;
      dist[n.x + dimx * n.y] = dist[c.x + dimx * c.y] + 1;
      q.push_back(n);
    }
  }
  return dist;
}

class Environment {  // cbs_ta.cpp:250-520, the parts a low-level search and a fixed-assignment conflict tree touch
 public:
  Environment(int dimx, int dimy, std::unordered_set<Cell, CellHash> obstacles)
      : m_dimx(dimx), m_dimy(dimy), m_obstacles(std::move(obstacles)) {}

  // `task` == nullptr: the agent has no task.  `table`: shortestPathTable(…, *task) (ignored without a task).
  void setLowLevelContext(std::size_t agentIdx, const Constraints* constraints, const Cell* task,
                          const std::vector<int>* table) {  // :283-303
    m_agent = agentIdx;
    m_goal = task;
    m_table = table;
    m_constraints = constraints;
    m_lastGoalConstraint = -1;
    if (m_goal != nullptr) {
      for (const auto& vc : constraints->vertex)
        if (vc.x == m_goal->x && vc.y == m_goal->y) m_lastGoalConstraint = std::max(m_lastGoalConstraint, vc.time);
    } else {
      for (const auto& vc : constraints->vertex) m_lastGoalConstraint = std::max(m_lastGoalConstraint, vc.time);
    }
  }

  int admissibleHeuristic(const State& s) const {  // :305-311
    if (m_goal != nullptr) return (*m_table)[s.x + m_dimx * s.y];
    return 0;
  }

  bool isSolution(const State& s) const {  // :313-319
    bool atGoal = true;
    if (m_goal != nullptr) atGoal = s.x == m_goal->x && s.y == m_goal->y;
    return atGoal && s.time > m_lastGoalConstraint;
  }

  void getNeighbors(const State& s, std::vector<Neighbor<State, Action, int>>& out) const {  // :321-367
    out.clear();
    static const int dx[5] = {0, -1, 1, 0, 0};
    static const int dy[5] = {0, 0, 0, 1, -1};
    static const Action act[5] = {Action::Wait, Action::Left, Action::Right, Action::Up, Action::Down};
    for (int k = 0; k < 5; ++k) {
      State n(s.time + 1, s.x + dx[k], s.y + dy[k]);
      if (!(stateValid(n) && transitionValid(s, n))) continue;
      int cost = 1;
      if (k == 0) {  // Wait: free at the goal (and everywhere for an agent without a task)
        bool atGoal = true;
        if (m_goal != nullptr) atGoal = s.x == m_goal->x && s.y == m_goal->y;
        cost = atGoal ? 0 : 1;
      }
      out.emplace_back(n, act[k], cost);
    }
  }

  // cbs_ta.cpp:369-427 — like example/cbs.cpp's scan, except that max_t is the LONGEST PATH'S STATE COUNT (cbs.cpp: that
  // minus one, cbs.cpp:338-343), so the time step at which the longest path ends is checked here too
  bool getFirstConflict(const std::vector<Plan>& sol, Conflict& c) const {
    int maxT = 0;
    for (const auto& p : sol) maxT = std::max<int>(maxT, int(p.states.size()));
    for (int t = 0; t < maxT; ++t) {
      for (std::size_t i = 0; i < sol.size(); ++i) {
        State si = stateAt(i, sol, t);
        for (std::size_t j = i + 1; j < sol.size(); ++j)
          if (si.sameCell(stateAt(j, sol, t))) {
            c.time = t; c.agent1 = i; c.agent2 = j; c.type = Conflict::Vertex;
            c.x1 = si.x; c.y1 = si.y; c.x2 = 0; c.y2 = 0;
            return true;
          }
      }
      for (std::size_t i = 0; i < sol.size(); ++i) {
        State ia = stateAt(i, sol, t), ib = stateAt(i, sol, t + 1);
        for (std::size_t j = i + 1; j < sol.size(); ++j) {
          State ja = stateAt(j, sol, t), jb = stateAt(j, sol, t + 1);
          if (ia.sameCell(jb) && ib.sameCell(ja)) {
            c.time = t; c.agent1 = i; c.agent2 = j; c.type = Conflict::Edge;
            c.x1 = ia.x; c.y1 = ia.y; c.x2 = ib.x; c.y2 = ib.y;
            return true;
          }
        }
      }
    }
    return false;
  }
  // createConstraintsFromConflict :429-447 is example/cbs.cpp's (it only looks at the conflict)
  void createConstraintsFromConflict(const Conflict& c, std::map<std::size_t, Constraints>& out) const {
    mapf::Environment e(m_dimx, m_dimy, {}, {});
    e.createConstraintsFromConflict(c, out);
  }

  void onExpandLowLevelNode(const State&, int, int) {
    ++m_llExpanded;
    ++m_llExpandedThisSearch;
  }
  long lowLevelExpanded() const { return m_llExpanded; }
  long m_llExpandedThisSearch = 0;
  long m_hlExpanded = 0;

 private:
  State stateAt(std::size_t i, const std::vector<Plan>& sol, std::size_t t) const {  // getState :472-481
    if (t < sol[i].states.size()) return sol[i].states[t].first;
    return sol[i].states.back().first;
  }
  bool stateValid(const State& s) const {  // :483-489
    return s.x >= 0 && s.x < m_dimx && s.y >= 0 && s.y < m_dimy &&
           m_obstacles.find(Cell{s.x, s.y}) == m_obstacles.end() &&
           m_constraints->vertex.find(VertexConstraint{s.time, s.x, s.y}) == m_constraints->vertex.end();
  }
  bool transitionValid(const State& a, const State& b) const {  // :491-496
    return m_constraints->edge.find(EdgeConstraint{a.time, a.x, a.y, b.x, b.y}) == m_constraints->edge.end();
  }

  int m_dimx, m_dimy;
  std::unordered_set<Cell, CellHash> m_obstacles;
  std::size_t m_agent = 0;
  const Cell* m_goal = nullptr;
  const std::vector<int>* m_table = nullptr;
  const Constraints* m_constraints = nullptr;
  int m_lastGoalConstraint = -1;
  long m_llExpanded = 0;
};

// cbs_ta.hpp:263-302: the adapter the low-level search is instantiated on (+ the harness's expansion cap)
struct LLEnv {
  Environment& env;
  long cap;  // < 0: unlimited (expansions of THIS search)
  int admissibleHeuristic(const State& s) { return env.admissibleHeuristic(s); }
  bool isSolution(const State& s) { return env.isSolution(s); }
  void getNeighbors(const State& s, std::vector<Neighbor<State, Action, int>>& n) { env.getNeighbors(s, n); }
  void onExpandNode(const State& s, int f, int g) {
    env.onExpandLowLevelNode(s, f, g);
    if (cap >= 0 && env.m_llExpandedThisSearch > cap) throw mapf::CapExceeded();
  }
  void onDiscover(const State&, int, int) {}
};

struct LowLevelCall {  // recorder entry: one search with everything it saw and returned
  std::size_t agent;
  bool hasTask;
  Cell task;
  Constraints constraints;
  bool ok;
  Plan plan;
  long expanded;
};

// One low-level search of the task-assignment callers (cbs_ta.hpp:106-109,156-158,196-199).
inline bool lowLevelSearch(Environment& env, std::size_t agent, const Constraints& c, const Cell* task,
                           const std::unordered_set<Cell, CellHash>& obstacles, int dimx, int dimy, const State& start,
                           Plan& out, long cap = -1) {
  std::vector<int> table;
  if (task) table = shortestPathTable(dimx, dimy, obstacles, *task);
  env.setLowLevelContext(agent, &c, task, task ? &table : nullptr);
  env.m_llExpandedThisSearch = 0;
  LLEnv ll{env, cap};
  AStar<State, Action, int, LLEnv, StateHash> search(ll);
  return search.search(start, out);
}

// cbs_ta.hpp:85-215 for ONE fixed assignment (tasks[i] = nullptr: agent i has no task): best-first conflict tree by cost.
// Returns false if a root search fails or the tree is exhausted; `cost` = sum of the path costs of the solution.
inline bool cbsFixedTasks(int dimx, int dimy, const std::unordered_set<Cell, CellHash>& obstacles,
                          const std::vector<State>& starts, const std::vector<const Cell*>& tasks, std::vector<Plan>& solution,
                          int& cost, std::vector<LowLevelCall>* recorder = nullptr, long maxHighLevel = 100000) {
  Environment env(dimx, dimy, obstacles);
  struct Node {
    std::vector<Plan> solution;
    std::vector<Constraints> constraints;
    int cost = 0;
    int id = 0;
  };
  struct Worse {  // cbs_ta.hpp:237-243 HighLevelNode::operator<: higher cost is worse
    bool operator()(const Node& a, const Node& b) const { return a.cost > b.cost; }
  };
  auto search = [&](std::size_t i, const Constraints& c, Plan& out) {
    bool ok = lowLevelSearch(env, i, c, tasks[i], obstacles, dimx, dimy, starts[i], out);
    if (recorder)
      recorder->push_back(LowLevelCall{i, tasks[i] != nullptr, tasks[i] ? *tasks[i] : Cell{0, 0}, c, ok, out, env.m_llExpandedThisSearch});
    return ok;
  };
  Node start;
  const std::size_t n = starts.size();
  start.solution.resize(n);
  start.constraints.resize(n);
  for (std::size_t i = 0; i < n; ++i) {
    if (!search(i, start.constraints[i], start.solution[i])) return false;
    start.cost += start.solution[i].cost;
  }
  ORACLE_HEAP<Node, Worse> open;
  open.push(start);
  int id = 1;
  while (!open.empty()) {
    Node P = open.top();
    if (++env.m_hlExpanded > maxHighLevel) return false;
    open.pop();
    Conflict conflict;
    if (!env.getFirstConflict(P.solution, conflict)) {
      solution = P.solution;
      cost = P.cost;
      return true;
    }
    std::map<std::size_t, Constraints> cons;
    env.createConstraintsFromConflict(conflict, cons);
    for (const auto& c : cons) {
      const std::size_t i = c.first;
      Node child = P;
      child.id = id;
      child.constraints[i].add(c.second);
      child.cost -= child.solution[i].cost;
      const bool ok = search(i, child.constraints[i], child.solution[i]);
      child.cost += child.solution[i].cost;
      if (ok) open.push(child);
      ++id;
    }
  }
  return false;
}

}  // namespace ta
}  // namespace oracle
