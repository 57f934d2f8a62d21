// ORACLE — TEST INFRASTRUCTURE ONLY (see heap_restated.hpp header).
//
// CPU restatement of the grid MAPF domain and of the two high-level conflict-tree searches.
//   value types / Environment : example/ecbs.cpp:16-522 (== example/cbs.cpp:16-569 minus the focal parts)
//   CBS::search               : include/libMultiRobotPlanning/cbs.hpp:85-172, adapter :209-244
//   ECBS::search              : include/libMultiRobotPlanning/ecbs.hpp:109-288, adapter :365-416
// Reference quirks kept on purpose: conflict scans stop at t < max_t; neighbour order
// Wait,Left,Right,Up,Down; goal test needs time > lastGoalConstraint; HL focal bound uses bestCost;
// a failed child search leaves cost 0 / stale fmin and the child is dropped.
#pragma once
#include <cstdlib>
#include <functional>
#include <map>
#include <unordered_set>
#include <vector>

#include "search_restated.hpp"

namespace oracle {
namespace mapf {
#ifdef ORACLE_SEARCH_STATS
void searchStatsSink(const SearchStats& st);  // oracle_capi.cpp (diagnostic build)
#endif

inline void hashCombine(std::size_t& seed, std::size_t v) {  // boost::hash_combine formula
  seed ^= v + 0x9e3779b9 + (seed << 6) + (seed >> 2);
}

struct State {  // ecbs.cpp:16-33 — equality includes time
  State(int time = 0, int x = 0, int y = 0) : time(time), x(x), y(y) {}
  bool operator==(const State& o) const { return time == o.time && x == o.x && y == o.y; }
  bool sameCell(const State& o) const { return x == o.x && y == o.y; }
  int time, x, y;
};
struct StateHash {
  std::size_t operator()(const State& s) const {
    std::size_t seed = 0;
    hashCombine(seed, std::hash<int>()(s.time));
    hashCombine(seed, std::hash<int>()(s.x));
    hashCombine(seed, std::hash<int>()(s.y));
    return seed;
  }
};

enum class Action { Up, Down, Left, Right, Wait };  // ecbs.cpp:49-55 (enum values 0..4)

struct Cell {
  int x, y;
  bool operator==(const Cell& o) const { return x == o.x && y == o.y; }
};
struct CellHash {
  std::size_t operator()(const Cell& c) const {
    std::size_t seed = 0;
    hashCombine(seed, std::hash<int>()(c.x));
    hashCombine(seed, std::hash<int>()(c.y));
    return seed;
  }
};

struct VertexConstraint {  // ecbs.cpp:108-125
  int time, x, y;
  bool operator==(const VertexConstraint& o) const { return time == o.time && x == o.x && y == o.y; }
};
struct EdgeConstraint {  // ecbs.cpp:140-163
  int time, x1, y1, x2, y2;
  bool operator==(const EdgeConstraint& o) const {
    return time == o.time && x1 == o.x1 && y1 == o.y1 && x2 == o.x2 && y2 == o.y2;
  }
};
struct VCHash {
  std::size_t operator()(const VertexConstraint& c) const {
    std::size_t seed = 0;
    hashCombine(seed, std::hash<int>()(c.time));
    hashCombine(seed, std::hash<int>()(c.x));
    hashCombine(seed, std::hash<int>()(c.y));
    return seed;
  }
};
struct ECHash {
  std::size_t operator()(const EdgeConstraint& c) const {
    std::size_t seed = 0;
    hashCombine(seed, std::hash<int>()(c.time));
    hashCombine(seed, std::hash<int>()(c.x1));
    hashCombine(seed, std::hash<int>()(c.y1));
    hashCombine(seed, std::hash<int>()(c.x2));
    hashCombine(seed, std::hash<int>()(c.y2));
    return seed;
  }
};

struct Constraints {  // ecbs.cpp:180-214
  std::unordered_set<VertexConstraint, VCHash> vertex;
  std::unordered_set<EdgeConstraint, ECHash> edge;
  void add(const Constraints& o) {
    vertex.insert(o.vertex.begin(), o.vertex.end());
    edge.insert(o.edge.begin(), o.edge.end());
  }
};

struct Conflict {  // ecbs.cpp:80-106
  enum Type { Vertex, Edge };
  int time;
  std::size_t agent1, agent2;
  Type type;
  int x1, y1, x2, y2;
};

typedef PlanResult<State, Action, int> Plan;

class Environment {  // ecbs.cpp:247-522
 public:
  Environment(int dimx, int dimy, std::unordered_set<Cell, CellHash> obstacles, std::vector<Cell> goals)
      : m_dimx(dimx), m_dimy(dimy), m_obstacles(std::move(obstacles)), m_goals(std::move(goals)) {}

  void setLowLevelContext(std::size_t agentIdx, const Constraints* constraints) {  // :264-274
    m_agent = agentIdx;
    m_constraints = constraints;
    m_lastGoalConstraint = -1;
    for (const auto& vc : constraints->vertex)
      if (vc.x == m_goals[m_agent].x && vc.y == m_goals[m_agent].y)
        m_lastGoalConstraint = std::max(m_lastGoalConstraint, vc.time);
  }

  int admissibleHeuristic(const State& s) const {  // :276-279
    return std::abs(s.x - m_goals[m_agent].x) + std::abs(s.y - m_goals[m_agent].y);
  }

  int focalStateHeuristic(const State& s, int, const std::vector<Plan>& sol) const {  // :282-295
    int n = 0;
    for (std::size_t i = 0; i < sol.size(); ++i)
      if (i != m_agent && !sol[i].states.empty() && s.sameCell(stateAt(i, sol, s.time))) ++n;
    return n;
  }

  int focalTransitionHeuristic(const State& a, const State& b, int, int,
                               const std::vector<Plan>& sol) const {  // :298-312
    int n = 0;
    for (std::size_t i = 0; i < sol.size(); ++i)
      if (i != m_agent && !sol[i].states.empty()) {
        State oa = stateAt(i, sol, a.time);
        State ob = stateAt(i, sol, b.time);
        if (a.sameCell(ob) && b.sameCell(oa)) ++n;
      }
    return n;
  }

  int focalHeuristic(const std::vector<Plan>& sol) const {  // :315-350
    int n = 0;
    int maxT = 0;
    for (const auto& p : sol) maxT = std::max<int>(maxT, int(p.states.size()) - 1);
    for (int t = 0; t < maxT; ++t) {
      for (std::size_t i = 0; i < sol.size(); ++i) {
        State si = stateAt(i, sol, t);
        for (std::size_t j = i + 1; j < sol.size(); ++j)
          if (si.sameCell(stateAt(j, sol, t))) ++n;
      }
      for (std::size_t i = 0; i < sol.size(); ++i) {
        State ia = stateAt(i, sol, t), ib = stateAt(i, sol, t + 1);
        for (std::size_t j = i + 1; j < sol.size(); ++j) {
          State ja = stateAt(j, sol, t), jb = stateAt(j, sol, t + 1);
          if (ia.sameCell(jb) && ib.sameCell(ja)) ++n;
        }
      }
    }
    return n;
  }

  bool isSolution(const State& s) const {  // :352-355
    return s.x == m_goals[m_agent].x && s.y == m_goals[m_agent].y && s.time > m_lastGoalConstraint;
  }

  void getNeighbors(const State& s, std::vector<Neighbor<State, Action, int>>& out) const {  // :357-399
    out.clear();
    static const int dx[5] = {0, -1, 1, 0, 0};
    static const int dy[5] = {0, 0, 0, 1, -1};
    static const Action act[5] = {Action::Wait, Action::Left, Action::Right, Action::Up, Action::Down};
    for (int k = 0; k < 5; ++k) {
      State n(s.time + 1, s.x + dx[k], s.y + dy[k]);
      if (stateValid(n) && transitionValid(s, n)) out.emplace_back(n, act[k], 1);
    }
  }

  bool getFirstConflict(const std::vector<Plan>& sol, Conflict& c) const {  // :401-452
    int maxT = 0;
    for (const auto& p : sol) maxT = std::max<int>(maxT, int(p.states.size()) - 1);
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

  void createConstraintsFromConflict(const Conflict& c, std::map<std::size_t, Constraints>& out) const {  // :454-472
    if (c.type == Conflict::Vertex) {
      Constraints k;
      k.vertex.insert(VertexConstraint{c.time, c.x1, c.y1});
      out[c.agent1] = k;
      out[c.agent2] = k;
    } else {
      Constraints k1, k2;
      k1.edge.insert(EdgeConstraint{c.time, c.x1, c.y1, c.x2, c.y2});
      k2.edge.insert(EdgeConstraint{c.time, c.x2, c.y2, c.x1, c.y1});
      out[c.agent1] = k1;
      out[c.agent2] = k2;
    }
  }

  void onExpandHighLevelNode(int) { ++m_hlExpanded; }
  void onExpandLowLevelNode(const State&, int, int) {
    ++m_llExpanded;
    ++m_llExpandedThisSearch;
  }
  long highLevelExpanded() const { return m_hlExpanded; }
  long lowLevelExpanded() const { return m_llExpanded; }

  // harness additions (the reference has no limits): expansion cap for one LL search
  long m_llExpandedThisSearch = 0;

 private:
  State stateAt(std::size_t i, const std::vector<Plan>& sol, std::size_t t) const {  // :486-495
    if (t < sol[i].states.size()) return sol[i].states[t].first;
    return sol[i].states.back().first;
  }
  bool stateValid(const State& s) const {  // :497-503
    return s.x >= 0 && s.x < m_dimx && s.y >= 0 && s.y < m_dimy &&
           m_obstacles.find(Cell{s.x, s.y}) == m_obstacles.end() &&
           m_constraints->vertex.find(VertexConstraint{s.time, s.x, s.y}) == m_constraints->vertex.end();
  }
  bool transitionValid(const State& a, const State& b) const {  // :505-510
    return m_constraints->edge.find(EdgeConstraint{a.time, a.x, a.y, b.x, b.y}) == m_constraints->edge.end();
  }

  int m_dimx, m_dimy;
  std::unordered_set<Cell, CellHash> m_obstacles;
  std::vector<Cell> m_goals;
  std::size_t m_agent = 0;
  const Constraints* m_constraints = nullptr;
  int m_lastGoalConstraint = -1;
  long m_hlExpanded = 0;
  long m_llExpanded = 0;
};

// Thrown by the adapters when a harness-imposed cap is exceeded (never happens in the reference, which
// would simply keep running).
struct CapExceeded {};

struct Limits {
  long maxLowLevelExpansionsPerSearch = -1;  // <0: unlimited
  long maxLowLevelExpansionsTotal = -1;
  long maxHighLevelExpansions = -1;
};

// Optional recorder of every low-level call (used to harvest kernel parity cases).
struct LowLevelCall {
  std::size_t agent;
  Constraints constraints;
  std::vector<Plan> solutionContext;  // ECBS only (as seen by the focal heuristics); empty for CBS
  bool success;
  Plan result;
  long expanded;
};

// ---------------------------------------------------------------------------------------------------
class CBS {  // cbs.hpp:79-249
  struct HLNode {
    std::vector<Plan> solution;
    std::vector<Constraints> constraints;
    int cost;
    int id;
  };
  struct HLLess {  // cbs.hpp:187-191
    bool operator()(const HLNode& a, const HLNode& b) const { return a.cost > b.cost; }
  };
  struct LLEnv {  // cbs.hpp:209-244
    LLEnv(Environment& env, std::size_t agent, const Constraints& c, const Limits& lim) : m_env(env), m_lim(lim) {
      m_env.setLowLevelContext(agent, &c);
      m_env.m_llExpandedThisSearch = 0;
    }
    int admissibleHeuristic(const State& s) { return m_env.admissibleHeuristic(s); }
    bool isSolution(const State& s) { return m_env.isSolution(s); }
    void getNeighbors(const State& s, std::vector<Neighbor<State, Action, int>>& n) { m_env.getNeighbors(s, n); }
    void onExpandNode(const State& s, int f, int g) {
      m_env.onExpandLowLevelNode(s, f, g);
      if (m_lim.maxLowLevelExpansionsPerSearch >= 0 &&
          m_env.m_llExpandedThisSearch > m_lim.maxLowLevelExpansionsPerSearch)
        throw CapExceeded();
      if (m_lim.maxLowLevelExpansionsTotal >= 0 && m_env.lowLevelExpanded() > m_lim.maxLowLevelExpansionsTotal)
        throw CapExceeded();
    }
    void onDiscover(const State&, int, int) {}
    Environment& m_env;
    const Limits& m_lim;
  };

 public:
  CBS(Environment& env, Limits lim = Limits()) : m_env(env), m_lim(lim) {}
  std::vector<LowLevelCall>* recorder = nullptr;

  bool search(const std::vector<State>& starts, std::vector<Plan>& solution) {
    HLNode root;
    root.solution.resize(starts.size());
    root.constraints.resize(starts.size());
    root.cost = 0;
    root.id = 0;
    for (std::size_t i = 0; i < starts.size(); ++i) {
      if (!lowLevel(i, root.constraints[i], starts[i], root.solution[i])) return false;
      root.cost += root.solution[i].cost;
    }
    ORACLE_HEAP<HLNode, HLLess> open;
    open.push(root);
    solution.clear();
    int id = 1;
    while (!open.empty()) {
      HLNode P = open.top();
      m_env.onExpandHighLevelNode(P.cost);
      if (m_lim.maxHighLevelExpansions >= 0 && m_env.highLevelExpanded() > m_lim.maxHighLevelExpansions)
        throw CapExceeded();
      open.pop();

      Conflict conflict;
      if (!m_env.getFirstConflict(P.solution, conflict)) {
        solution = P.solution;
        return true;
      }
      std::map<std::size_t, Constraints> split;
      m_env.createConstraintsFromConflict(conflict, split);
      for (const auto& c : split) {
        std::size_t i = c.first;
        HLNode child = P;
        child.id = id;
        child.constraints[i].add(c.second);
        child.cost -= child.solution[i].cost;
        bool ok = lowLevel(i, child.constraints[i], starts[i], child.solution[i]);
        child.cost += child.solution[i].cost;
        if (ok) open.push(child);
        ++id;
      }
    }
    return false;
  }

 private:
  bool lowLevel(std::size_t agent, const Constraints& c, const State& start, Plan& out) {
    LLEnv llenv(m_env, agent, c, m_lim);
    AStar<State, Action, int, LLEnv, StateHash> ll(llenv);
    bool ok = ll.search(start, out);
    if (recorder) recorder->push_back(LowLevelCall{agent, c, {}, ok, out, m_env.m_llExpandedThisSearch});
    return ok;
  }
  Environment& m_env;
  Limits m_lim;
};

// ---------------------------------------------------------------------------------------------------
class ECBS {  // ecbs.hpp:103-423
  struct HLNode {
    std::vector<Plan> solution;
    std::vector<Constraints> constraints;
    int cost;
    int LB;
    int focalHeuristic;
    int id;
  };
  struct HLLess {  // ecbs.hpp:321-325
    bool operator()(const HLNode& a, const HLNode& b) const { return a.cost > b.cost; }
  };
  typedef ORACLE_HEAP<HLNode, HLLess> Open;
  typedef Open::handle_type OpenHandle;
  struct FocalLess {  // ecbs.hpp:344-352
    const Open* open;
    bool operator()(const OpenHandle& h1, const OpenHandle& h2) const {
      const HLNode& a = (*open)[h1];
      const HLNode& b = (*open)[h2];
      if (a.focalHeuristic != b.focalHeuristic) return a.focalHeuristic > b.focalHeuristic;
      return a.cost > b.cost;
    }
  };
  struct LLEnv {  // ecbs.hpp:365-416
    LLEnv(Environment& env, std::size_t agent, const Constraints& c, const std::vector<Plan>& sol, const Limits& lim)
        : m_env(env), m_sol(sol), m_lim(lim) {
      m_env.setLowLevelContext(agent, &c);
      m_env.m_llExpandedThisSearch = 0;
    }
    int admissibleHeuristic(const State& s) { return m_env.admissibleHeuristic(s); }
    int focalStateHeuristic(const State& s, int g) { return m_env.focalStateHeuristic(s, g, m_sol); }
    int focalTransitionHeuristic(const State& a, const State& b, int ga, int gb) {
      return m_env.focalTransitionHeuristic(a, b, ga, gb, m_sol);
    }
    bool isSolution(const State& s) { return m_env.isSolution(s); }
    void getNeighbors(const State& s, std::vector<Neighbor<State, Action, int>>& n) { m_env.getNeighbors(s, n); }
    void onExpandNode(const State& s, int f, int g) {
      m_env.onExpandLowLevelNode(s, f, g);
      if (m_lim.maxLowLevelExpansionsPerSearch >= 0 &&
          m_env.m_llExpandedThisSearch > m_lim.maxLowLevelExpansionsPerSearch)
        throw CapExceeded();
      if (m_lim.maxLowLevelExpansionsTotal >= 0 && m_env.lowLevelExpanded() > m_lim.maxLowLevelExpansionsTotal)
        throw CapExceeded();
    }
    void onDiscover(const State&, int, int) {}
    Environment& m_env;
    const std::vector<Plan>& m_sol;
    const Limits& m_lim;
  };

 public:
  ECBS(Environment& env, float w, Limits lim = Limits()) : m_env(env), m_w(w), m_lim(lim) {}
  std::vector<LowLevelCall>* recorder = nullptr;

  bool search(const std::vector<State>& starts, std::vector<Plan>& solution) {
    HLNode root;
    root.solution.resize(starts.size());
    root.constraints.resize(starts.size());
    root.cost = 0;
    root.LB = 0;
    root.id = 0;
    for (std::size_t i = 0; i < starts.size(); ++i) {
      // (warm start branch ecbs.hpp:119-124 is never taken by example/ecbs.cpp:578 — omitted)
      if (!lowLevel(i, root.constraints[i], root.solution, starts[i], root.solution[i])) return false;
      root.cost += root.solution[i].cost;
      root.LB += root.solution[i].fmin;
    }
    root.focalHeuristic = m_env.focalHeuristic(root.solution);

    Open open;
    ORACLE_HEAP<OpenHandle, FocalLess> focal(FocalLess{&open});
    OpenHandle h0 = open.push(root);
    focal.push(h0);
    int bestCost = open[h0].cost;

    solution.clear();
    int id = 1;
    while (!open.empty()) {
      {  // ecbs.hpp:170-190 — bound is bestCost * w (float), not LB * w
        int oldBest = bestCost;
        bestCost = open.top().cost;
        if (bestCost > oldBest) {
          open.orderedWalk([&](OpenHandle h) {
            int val = open[h].cost;
            if (val > oldBest * m_w && val <= bestCost * m_w) focal.push(h);
            if (val > bestCost * m_w) return false;
            return true;
          });
        }
      }
      OpenHandle h = focal.top();
      HLNode P = open[h];
      m_env.onExpandHighLevelNode(P.cost);
      if (m_lim.maxHighLevelExpansions >= 0 && m_env.highLevelExpanded() > m_lim.maxHighLevelExpansions)
        throw CapExceeded();
      focal.pop();
      open.erase(h);

      Conflict conflict;
      if (!m_env.getFirstConflict(P.solution, conflict)) {
        solution = P.solution;
        return true;
      }
      std::map<std::size_t, Constraints> split;
      m_env.createConstraintsFromConflict(conflict, split);
      for (const auto& c : split) {
        std::size_t i = c.first;
        HLNode child = P;
        child.id = id;
        child.constraints[i].add(c.second);
        child.cost -= child.solution[i].cost;
        child.LB -= child.solution[i].fmin;
        bool ok = lowLevel(i, child.constraints[i], child.solution, starts[i], child.solution[i]);
        child.cost += child.solution[i].cost;
        child.LB += child.solution[i].fmin;
        child.focalHeuristic = m_env.focalHeuristic(child.solution);
        if (ok) {
          OpenHandle hc = open.push(child);
          if (child.cost <= bestCost * m_w) focal.push(hc);
        }
        ++id;
      }
    }
    return false;
  }

 private:
  bool lowLevel(std::size_t agent, const Constraints& c, const std::vector<Plan>& context, const State& start,
                Plan& out) {
    std::vector<Plan> ctxCopy;
    if (recorder) ctxCopy = context;
    LLEnv llenv(m_env, agent, c, context, m_lim);
    AStarEpsilon<State, Action, int, LLEnv, StateHash> ll(llenv, m_w);
#ifdef ORACLE_SEARCH_STATS
    SearchStats st;
    g_stats = &st;
#endif
    bool ok = ll.search(start, out);
#ifdef ORACLE_SEARCH_STATS
    g_stats = nullptr;
    searchStatsSink(st);
#endif
    if (recorder) recorder->push_back(LowLevelCall{agent, c, ctxCopy, ok, out, m_env.m_llExpandedThisSearch});
    return ok;
  }
  Environment& m_env;
  float m_w;
  Limits m_lim;
};

}  // namespace mapf
}  // namespace oracle
